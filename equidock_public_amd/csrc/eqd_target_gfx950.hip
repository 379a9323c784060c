// Identifies the real gfx950 build of the C ABI (tests/hostsim links its own definition returning 1).
extern "C" int eqd_is_simulator(void) { return 0; }
