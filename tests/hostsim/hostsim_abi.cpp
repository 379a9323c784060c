// TEST INFRASTRUCTURE: marks the x86 simulator build of the C ABI.
extern "C" int eqd_is_simulator(void) { return 1; }
