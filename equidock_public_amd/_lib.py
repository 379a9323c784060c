"""ctypes binding of libequidock_hip.so (the C ABI declared in include/equidock_hip.h).

There is NO fallback: if the HIP library is missing or a tensor is not on the GPU the calls
raise.  (tests/hostsim builds an x86 simulator of the same ABI to debug index arithmetic in a
container without a GPU; it is only ever loaded explicitly through
`load_library_for_testing`, and `eqd_is_simulator()` tells the two apart.)
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libequidock_hip.so')
if os.environ.get('EQD_EXP_LIBRARY'):      # knock-out builds of the SAME sources (profiles/exp_r06_knockouts.sh: timing experiments)
    LIB_PATH = os.environ['EQD_EXP_LIBRARY']

EQD_MAX_SRC = 6
ABI_VERSION = 9
PARAMS_PER_LAYER = 19
GLOBAL_PARAMS = 5

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class EqdGraph(C.Structure):
    _fields_ = [('n_pairs', C.c_int32), ('n_lig', C.c_int32), ('n_rec', C.c_int32), ('n_nodes', C.c_int32),
                ('n_edges', C.c_int32), ('n_tiles', C.c_int32), ('n_att_items', C.c_int32), ('max_seg', C.c_int32),
                ('seg_off', C.c_void_p), ('src', C.c_void_p), ('dst', C.c_void_p), ('rowptr', C.c_void_p),
                ('csc_ptr', C.c_void_p), ('csc_eid', C.c_void_p), ('tile_node', C.c_void_p),
                ('att_items', C.c_void_p), ('res_id', C.c_void_p), ('mu_r_norm', C.c_void_p), ('he', C.c_void_p),
                ('x0', C.c_void_p), ('he_bf16', C.c_void_p)]


class EqdModelDesc(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('d_emb', C.c_int32), ('d_hid', C.c_int32),
                ('use_mean_node_features', C.c_int32), ('edge_feats', C.c_int32), ('n_heads', C.c_int32),
                ('cross_msgs', C.c_int32), ('use_dist_in_layers', C.c_int32), ('use_edge_features', C.c_int32),
                ('skip_weight_h', C.c_float), ('x_connection_init', C.c_float), ('lrelu_slope', C.c_float),
                ('ln_eps', C.c_float), ('svd_seed', C.c_int32), ('storage_bf16', C.c_int32)]


class EqdDropout(C.Structure):
    _fields_ = [('p', C.c_float), ('edge_z1', C.c_void_p), ('edge_ch', C.c_void_p), ('node', C.c_void_p),
                ('head', C.c_void_p)]


class EqdLinSrc(C.Structure):
    _fields_ = [('X', C.c_void_p), ('mask', C.c_void_p), ('W', C.c_void_p), ('ldx', C.c_int32), ('K', C.c_int32),
                ('w_rs', C.c_int32), ('w_cs', C.c_int32)]


class EqdLinJob(C.Structure):
    _fields_ = [('s', EqdLinSrc * EQD_MAX_SRC), ('nsrc', C.c_int32), ('M', C.c_int32), ('act', C.c_int32),
                ('rows', C.c_int32), ('bias', C.c_void_p), ('ln_g', C.c_void_p), ('ln_b', C.c_void_p),
                ('pre_ln', C.c_void_p), ('ld_pre', C.c_int32), ('R', C.c_void_p), ('ldr', C.c_int32),
                ('alpha', C.c_float), ('beta', C.c_float), ('slope', C.c_float), ('ln_eps', C.c_float),
                ('Y', C.c_void_p), ('ldy', C.c_int32), ('bf16', C.c_int32), ('mul', C.c_void_p), ('ld_mul', C.c_int32), ('pad_to', C.c_int32),
                ('Yb', C.c_void_p), ('ldyb', C.c_int32)]


class EqdAtbJob(C.Structure):
    _fields_ = [('X', C.c_void_p), ('xmask', C.c_void_p), ('ldx', C.c_int32), ('M', C.c_int32), ('Y', C.c_void_p),
                ('ldy', C.c_int32), ('N', C.c_int32), ('rows', C.c_int32), ('out', C.c_void_p), ('o_rs', C.c_int32),
                ('o_cs', C.c_int32), ('bias_out', C.c_void_p), ('slope', C.c_float), ('scale', C.c_float), ('bf16', C.c_int32),
                ('y_bf16', C.c_int32)]


class EqdEdgeParams(C.Structure):
    _fields_ = [('W1', C.c_void_p), ('ldw1', C.c_int32), ('d_in', C.c_int32), ('ln_g', C.c_void_p),
                ('ln_b', C.c_void_p), ('W2', C.c_void_p), ('b2', C.c_void_p), ('Wc1', C.c_void_p),
                ('bc1', C.c_void_p), ('wc2', C.c_void_p), ('bc2', C.c_void_p), ('slope', C.c_float),
                ('ln_eps', C.c_float), ('eta', C.c_float), ('use_dist', C.c_int32), ('use_he', C.c_int32),
                ('bf16', C.c_int32), ('drop_z1', C.c_void_p), ('drop_ch', C.c_void_p), ('drop_scale', C.c_float),
                ('aggr_bf16', C.c_void_p), ('xh_save', C.c_void_p), ('rstd_save', C.c_void_p), ('zpos_save', C.c_void_p)]


class EqdNodeUpdateParams(C.Structure):
    _fields_ = [('d_in', C.c_int32), ('d0', C.c_int32), ('d_out', C.c_int32), ('ld_cross', C.c_int32), ('Wn1', C.c_void_p),
                ('bn1', C.c_void_p), ('ln_g', C.c_void_p), ('ln_b', C.c_void_p), ('Wn2', C.c_void_p), ('bn2', C.c_void_p),
                ('skip_weight_h', C.c_float), ('slope', C.c_float), ('ln_eps', C.c_float), ('bf16', C.c_int32),
                ('drop_mul', C.c_void_p)]


class EqdNodeUpdateGrads(C.Structure):
    _fields_ = [('dWn1', C.c_void_p), ('dbn1', C.c_void_p), ('dln_g', C.c_void_p), ('dln_b', C.c_void_p),
                ('dWn2', C.c_void_p), ('dbn2', C.c_void_p)]


class EqdEdgeGrads(C.Structure):
    _fields_ = [('dW1', C.c_void_p), ('ldw1', C.c_int32), ('db1_unused', C.c_void_p), ('dln_g', C.c_void_p),
                ('dln_b', C.c_void_p), ('dW2', C.c_void_p), ('db2', C.c_void_p), ('dWc1', C.c_void_p),
                ('dbc1', C.c_void_p), ('dwc2', C.c_void_p), ('dbc2', C.c_void_p)]


_lib = None
_is_sim = False
profiling = False    # set by bench.py around an eqd_profile_begin/_end bracket: model.py then marks the torch-side work


class EquidockHipError(RuntimeError):
    pass


def _declare(lib):
    lib.eqd_abi_version.restype = C.c_int
    lib.eqd_last_error.restype = C.c_char_p
    lib.eqd_tile_edges.restype = C.c_int
    lib.eqd_is_simulator.restype = C.c_int
    lib.eqd_model_saved_bytes.restype = C.c_size_t
    lib.eqd_model_scratch_bytes.restype = C.c_size_t
    lib.eqd_atb_partial_bytes.restype = C.c_size_t
    lib.eqd_edge_message_bwd_workspace_bytes.restype = C.c_size_t
    lib.eqd_keypoint_pool_bwd_workspace_bytes.restype = C.c_size_t
    lib.eqd_clash_workspace_bytes.restype = C.c_size_t
    lib.eqd_node_update_bwd_workspace_bytes.restype = C.c_size_t
    lib.eqd_cross_attention_bwd_ds_workspace_bytes.restype = C.c_size_t
    lib.eqd_profile_name.restype = C.c_char_p
    lib.eqd_profile_us.restype = C.c_float
    lib.eqd_tunables_reload.restype = None
    for name in ('eqd_model_layer_state', 'eqd_model_lrelu_signs', 'eqd_model_head_backward', 'eqd_profile_begin', 'eqd_profile_end', 'eqd_profile_mark', 'eqd_ctx_create', 'eqd_ctx_destroy', 'eqd_model_check', 'eqd_model_forward', 'eqd_model_backward', 'eqd_linear', 'eqd_atb',
                 'eqd_edge_message_fwd', 'eqd_edge_message_bwd', 'eqd_edge_message_bwd_kernel_only',
                 'eqd_cross_attention_fwd', 'eqd_cross_attention_fwd_bf16', 'eqd_cross_attention_bwd_bf16',
                 'eqd_cross_attention_bwd', 'eqd_cross_attention_bwd_ds',
           'eqd_keypoint_pool_fwd', 'eqd_keypoint_pool_bwd', 'eqd_kabsch_fwd', 'eqd_kabsch_bwd',
                 'eqd_rigid_apply_fwd', 'eqd_rigid_apply_bwd', 'eqd_pair_losses_fwd', 'eqd_pair_losses_bwd', 'eqd_scalar_loss', 'eqd_pocket_ot_cost',
                 'eqd_pocket_ot_fwd', 'eqd_pocket_ot_bwd', 'eqd_rigid_augment', 'eqd_protein_graph_distances',
                 'eqd_protein_graph_select', 'eqd_protein_graph_edges', 'eqd_clash_iterations', 'eqd_dropout_pack_edges', 'eqd_dropout_draw',
                 'eqd_node_update_fwd', 'eqd_node_update_bwd', 'eqd_selftest_lane_exchanges'):
        getattr(lib, name).restype = C.c_int


EXPORTS = ('eqd_model_layer_state', 'eqd_model_lrelu_signs', 'eqd_model_head_backward', 'eqd_profile_begin', 'eqd_profile_end', 'eqd_profile_mark', 'eqd_profile_name', 'eqd_profile_us', 'eqd_ctx_create', 'eqd_ctx_destroy', 'eqd_abi_version', 'eqd_last_error', 'eqd_tile_edges', 'eqd_is_simulator', 'eqd_model_saved_bytes', 'eqd_model_saved_layout',
           'eqd_model_scratch_bytes', 'eqd_model_check', 'eqd_model_forward', 'eqd_model_backward', 'eqd_linear',
           'eqd_atb_partial_bytes', 'eqd_atb', 'eqd_edge_message_fwd', 'eqd_edge_message_bwd_workspace_bytes',
           'eqd_edge_message_bwd', 'eqd_edge_message_bwd_kernel_only', 'eqd_cross_attention_fwd',
           'eqd_cross_attention_bwd', 'eqd_cross_attention_bwd_ds', 'eqd_cross_attention_bwd_ds_workspace_bytes', 'eqd_cross_attention_fwd_bf16', 'eqd_cross_attention_bwd_bf16',
           'eqd_keypoint_pool_fwd', 'eqd_keypoint_pool_bwd',
           'eqd_keypoint_pool_bwd_workspace_bytes',
           'eqd_kabsch_fwd', 'eqd_kabsch_bwd', 'eqd_rigid_apply_fwd', 'eqd_rigid_apply_bwd', 'eqd_pair_losses_fwd',
           'eqd_pair_losses_bwd', 'eqd_scalar_loss', 'eqd_pocket_ot_cost', 'eqd_pocket_ot_fwd', 'eqd_pocket_ot_bwd',
           'eqd_rigid_augment', 'eqd_protein_graph_distances', 'eqd_protein_graph_select', 'eqd_protein_graph_edges',
           'eqd_clash_workspace_bytes', 'eqd_clash_iterations', 'eqd_dropout_pack_edges', 'eqd_dropout_draw',
           'eqd_tunables_reload', 'eqd_node_update_fwd', 'eqd_node_update_bwd_workspace_bytes', 'eqd_node_update_bwd',
           'eqd_selftest_lane_exchanges')


def load_library():
    """Load the gfx950 library built by equidock_public_amd/build.py. Raises if it is missing."""
    global _lib, _is_sim
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EquidockHipError(
            f"{LIB_PATH} is missing: build it with `python -m equidock_public_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the IEGMN hot path.")
    lib = C.CDLL(LIB_PATH)
    _declare(lib)
    if lib.eqd_abi_version() != ABI_VERSION:
        raise EquidockHipError(f"ABI version mismatch: library {lib.eqd_abi_version()} != {ABI_VERSION}")
    _lib, _is_sim = lib, bool(lib.eqd_is_simulator())
    return _lib


def load_library_for_testing(path):
    """TESTS ONLY: bind an explicitly given build of the ABI (the x86 simulator of tests/hostsim)."""
    global _lib, _is_sim
    lib = C.CDLL(path)
    _declare(lib)
    _lib, _is_sim = lib, bool(lib.eqd_is_simulator())
    return _lib


def unload_for_testing():
    global _lib, _is_sim
    _lib, _is_sim = None, False


tunables_generation = 0      # bumped by reload_tunables(): sizes cached per PackedGraph depend on the switches


def reload_tunables():
    """Make the loaded library re-read its EQD_* experiment switches (it snapshots them once per process; tests and A/B
    measurements that change the environment afterwards call this).  No-op when no library is loaded yet."""
    global tunables_generation
    tunables_generation += 1
    if _lib is not None:
        _lib.eqd_tunables_reload()


def is_simulator():
    return _is_sim


def check(rc):
    if rc != 0:
        raise EquidockHipError(f"libequidock_hip error {rc}: {_lib.eqd_last_error().decode()}")


def require_device(t, what='tensor'):
    """Every buffer handed to the library must live in HBM (or host memory for the simulator)."""
    if _is_sim:
        if t.is_cuda:
            raise EquidockHipError(f"{what}: the host simulator only takes CPU tensors")
    elif not t.is_cuda:
        raise EquidockHipError(
            f"{what} is on {t.device}: the IEGMN hot path runs only on an MI355X through libequidock_hip.so "
            "(no CPU fallback)")
    return t


class device_guard:
    """`with device_guard(dev):` - the tensors' GPU is the current HIP device while the C library enqueues work (it
    launches on the passed stream, but hipMemsetAsync / kernel launches resolve against the current device)."""

    def __init__(self, device):
        self._cm = None if (_is_sim or torch.device(device).type != 'cuda') else torch.cuda.device(device)

    def __enter__(self):
        if self._cm is not None:
            self._cm.__enter__()

    def __exit__(self, *exc):
        if self._cm is not None:
            self._cm.__exit__(*exc)
        return False


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(device):
    if _is_sim:
        return C.c_void_p(0)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_ctx = {}


def exec_ctx(device):
    """Execution context of the C library (an auxiliary stream on which the forward's attention kernel runs beside the
    edge-message kernel).  Off by default: with the current kernels one stream measured faster at every workload
    (B 4 274 vs 4 128 pairs/s, C 5 894 vs 5 810, E 440 vs 433) - the event fork/join costs more than the overlap
    gains.  EQD_FORK=1 turns it on."""
    if os.environ.get('EQD_FORK') != '1':
        return C.c_void_p(0)
    key = (id(_lib), str(device))
    if key not in _ctx:
        h = C.c_void_p(0)
        if not _is_sim:
            with torch.cuda.device(device):
                check(_lib.eqd_ctx_create(C.byref(h)))
        else:
            check(_lib.eqd_ctx_create(C.byref(h)))
        _ctx[key] = h
    return _ctx[key]


def graph_struct(p):
    """PackedGraph -> EqdGraph (pointers into the packed tensors, which the caller keeps alive)."""
    g = EqdGraph()
    g.n_pairs, g.n_lig, g.n_rec, g.n_nodes = p.n_pairs, p.n_lig, p.n_rec, p.n_nodes
    g.n_edges, g.n_tiles, g.n_att_items, g.max_seg = p.n_edges, p.n_tiles, p.n_att_items, p.max_seg
    for name in ('seg_off', 'src', 'dst', 'rowptr', 'csc_ptr', 'csc_eid', 'tile_node', 'att_items', 'res_id',
                 'mu_r_norm', 'he', 'x0', 'he_bf16'):
        t = getattr(p, name)
        require_device(t, 'graph.' + name)
        setattr(g, name, t.data_ptr())
    return g
