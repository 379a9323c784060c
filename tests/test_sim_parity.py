"""CPU tests: the product's kernel sources compiled for x86 with the host simulator
(tests/hostsim) and driven through the same C ABI + Python modules as on the GPU, compared with
torch restatements / the golden vectors.  This debugs index arithmetic without a GPU; the real
parity gate is tests/test_gpu_parity.py (-m gpu)."""
import pytest
import torch

from equidock_public_amd import _lib
from tests import parity_common as pc

DEV = torch.device('cpu')


@pytest.fixture(scope='module', autouse=True)
def simulator():
    from tests.hostsim import build as hs
    lib = hs.build()
    _lib.load_library_for_testing(lib)
    assert _lib.is_simulator()
    yield
    _lib.unload_for_testing()


def test_linear():
    pc.check_linear(DEV)


def test_linear_simple_form_is_bit_identical(monkeypatch):
    pc.check_linear_simple_form(DEV, monkeypatch)


def test_linear_simple80_form_is_bit_identical(monkeypatch):
    pc.check_linear_simple80_form(DEV, monkeypatch)


def test_training_step_is_bit_reproducible():
    pc.check_run_to_run_bits(DEV, cases=((True, 0.25, 1, 2, 40), (False, 0.25, 1, 2, 40)), runs=2)      # (host logic; the GPU test runs the sizes that matter)


def test_atb():
    pc.check_atb(DEV)


def test_edge_message_fwd_bwd():
    pc.check_edge(DEV)


def test_edge_message_bf16():
    pc.check_edge_bf16(DEV)


def test_edge_message_with_dropout_masks():
    pc.check_edge(DEV, drop=True)
    pc.check_edge(DEV, drop=True, bf16=True)


def test_dropout_training_through_the_kernels():
    pc.check_dropout_training(DEV)


def test_dropout_masks_drawn_by_the_library():
    pc.check_dropout_library(DEV)


@pytest.mark.parametrize('d', [64, 69, 80])
def test_cross_attention(d):
    pc.check_attention(DEV, d)


@pytest.mark.parametrize('split', ['0', '1'])
def test_attention_half_blocks(split, monkeypatch):
    """attention with one or two workgroups per 32-row work item (EQD_ATT_SPLIT forward, EQD_ATT_BWD_SPLIT backward)
    on the float4 paths, and a whole model without the splits (the defaults use them)"""
    monkeypatch.setenv('EQD_ATT_SPLIT', split)
    monkeypatch.setenv('EQD_ATT_BWD_SPLIT', split)
    pc.check_attention(DEV, 64)
    pc.check_attention(DEV, 80, sizes=((300, 257), (129, 64)))
    if split == '0':
        pc.check_model_case(DEV, 'D_degraded3')


def test_kabsch():
    pc.check_kabsch(DEV)


def test_keypoints_and_apply():
    pc.check_keypoints_and_apply(DEV)


@pytest.mark.parametrize('K', [3, 16, 37, 64, 70])
def test_keypoint_head_counts(K, monkeypatch):
    """num_att_heads other than 50: fewer heads than a lane group holds, whole 16-head blocks, a ragged last block, the
    largest count of the product backward (64) and one beyond it (the backward falls back to the one-head kernels)"""
    monkeypatch.setenv('EQD_KEYPOINT_MM', '1')
    pc.check_keypoints_and_apply(DEV, K=K)


@pytest.mark.parametrize('form', ['first', 'mm', 'mm_chunks', 'mm_long'])
def test_keypoint_kernel_forms(form, monkeypatch):
    """Keypoint pooling has two sets of kernels: one workgroup per (segment, head) (EQD_KEYPOINT_MM=0) and the matrix-product
    forms (eqd_keypoint_mm_inl.h; the backward with one or several row chunks per segment, and with the per-head dot taken
    from the forward's Y - the model path - or from a pass over the rows - the operator entry point)."""
    monkeypatch.setenv('EQD_KEYPOINT_MM', '0' if form == 'first' else '1')
    if form == 'mm_long':      # one long segment: the forward's 16-wave workgroups, a backward of 8 chunks (some of them empty)
        pc.check_keypoints_and_apply(DEV, sizes=((1030, 40),))
        return
    if form == 'mm_chunks':
        monkeypatch.setenv('EQD_KEYPOINT_NC', '2')
        pc.check_keypoints_and_apply(DEV, sizes=((150, 130), (140, 161)))      # 581 nodes: two partial du blocks per segment fit
    else:
        pc.check_keypoints_and_apply(DEV)
    pc.check_model_case(DEV, 'D_degraded3')
    pc.check_head_backward(DEV, [(41, 57), (66, 38)], layers=3, what=f'keypoint kernels: {form}')
    if form == 'first':      # 18 pairs = 36 segments: the key / query maps' backward in two segment groups (partial sums); the
                             # product kernels' groups: test_many_pairs_split_head_backward
        pc.check_keypoints_and_apply(DEV, sizes=tuple((12 + i, 30 - i) for i in range(18)))
        pc.check_head_backward(DEV, [(12 + i, 30 - i) for i in range(18)], layers=2, what=f'keypoint kernels, 18 pairs: {form}')
    names = pc.launch_names_of_a_step(DEV, 'D_degraded3')
    assert ('k_keypoint_bwd' in names) == (form != 'first') and ('k_keypoint_bwd_a' in names) == (form == 'first'), sorted(set(names))


@pytest.mark.parametrize('name', ['A_b1_shared5', 'D_degraded3', 'E_svd_guard'])
def test_model_vs_golden(name):
    pc.check_model_case(DEV, name)


@pytest.mark.parametrize('name', ['D_degraded3'])
def test_model_two_row_tiles(name, monkeypatch):
    """the 32-rows-per-workgroup variants of the row kernels (EQD_ROW_TILES=2) on a small golden case"""
    monkeypatch.setenv('EQD_ROW_TILES', '2')
    monkeypatch.setenv('EQD_ROWWAVE', '0')
    pc.check_linear(DEV)
    pc.check_model_case(DEV, name)


@pytest.mark.parametrize('rowwave', ['0', '1', '2'])
def test_row_kernels_both_forms(rowwave, monkeypatch):
    """The node-level chains have two kernels: k_rowwave (one wave per 16-row tile, chain in registers; every chain of the
    layers >= 1) and k_rowchain / k_linear (four waves per tile, for layer 0's 69-wide jobs and under EQD_ROWWAVE=0).  Both
    must reproduce the golden vectors, and the switch must really select the kernel."""
    monkeypatch.setenv('EQD_ROWWAVE', rowwave)
    pc.check_linear(DEV)
    pc.check_model_case(DEV, 'D_degraded3')
    pc.check_dropout_library(DEV)           # the dropout factors in each form's epilogue and LayerNorm backward
    names = pc.launch_names_of_a_step(DEV, 'D_degraded3')
    assert ('k_rowwave' in names) == (rowwave == '1') and ('k_rowres' in names) == (rowwave == '2'), sorted(set(names))
    assert 'k_rowchain' in names            # layer 0 (and everything under EQD_ROWWAVE=0)


@pytest.mark.parametrize('tps', ['3', '11', '16'])
def test_rowres_tiles_per_workgroup(tps, monkeypatch):
    """k_rowres with several tiles per workgroup (one and two tile slots per wave, idle waves, ragged last workgroup)"""
    monkeypatch.setenv('EQD_ROWWAVE', '2')
    monkeypatch.setenv('EQD_ROWRES_TPS', tps)
    pc.check_linear(DEV)
    pc.check_linear_atb_bf16(DEV)
    pc.check_model_case(DEV, 'D_degraded3')
    pc.check_first_layer_rowres80(DEV, rows=16 * int(tps) + 5)      # (k_rowres80: one / two tile slots, a ragged last tile)


def test_model_bf16_on_resident_row_kernels(monkeypatch):
    """bf16 mode with every row chain on the LDS-resident-weights kernels (what large batches run): k_rowres for the 64-wide
    layers, k_rowres80 for the first layer's forward chain (carrying layer 1's projections) and projection group"""
    monkeypatch.setenv('EQD_ROWWAVE', '2')
    monkeypatch.setenv('EQD_ROWRES_TPS', '3')
    pc.check_model_bf16(DEV, 'D_degraded3')
    pc.check_model_bf16_states(DEV, [(60, 75), (90, 48), (120, 100)], layers=4, seed=5, pair_seed=7, faithful=True, what='sim, resident')
    names = pc.launch_names_of_a_step(DEV, 'D_degraded3')
    assert 'k_rowres' in names


def test_model_bf16_mode():
    pc.check_model_bf16(DEV, 'D_degraded3')


def test_model_forked_attention_stream(monkeypatch):
    monkeypatch.setenv('EQD_FORK', '1')
    pc.check_model_case(DEV, 'D_degraded3')


def test_first_layer_on_rowres80():
    pc.check_first_layer_rowres80(DEV, rows=77)


def test_rowres80_dropout():
    pc.check_rowres80_dropout(DEV)


def test_bf16_storage_operators():
    pc.check_bf16_storage_ops(DEV)


def test_bf16_storage_of_the_saved_state(monkeypatch):
    pc.check_bf16_storage_model(DEV, monkeypatch)


def test_lane_exchanges():
    pc.check_lane_exchanges(DEV)


def test_node_update_operator():
    pc.check_node_update(DEV, rows=77)


def test_standalone_layer_through_the_library():
    pc.check_standalone_layer(DEV)


def test_flat_grads():
    pc.check_flat_grads_equal_autograd(DEV)


def test_model_vs_oracle_ragged():
    pc.check_model_vs_oracle_ragged(DEV, check_grads=False)
    pc.check_model_vs_oracle_ragged(DEV, sizes=((4, 4), (17, 5), (33, 64)))


@pytest.mark.parametrize('over', [dict(cross_msgs=False), dict(use_dist_in_layers=False),
                                  dict(use_edge_features_in_gmn=False), dict(use_mean_node_features=False),
                                  dict(x_connection_init=0.25), dict(num_att_heads=13),
                                  dict(cross_msgs=False, use_mean_node_features=False, x_connection_init=0.25,
                                       shared_layers=True)],
                         ids=lambda o: '+'.join(f'{k}={v}' for k, v in o.items()))
def test_option_toggles_vs_oracle(over):
    """every `args` switch the HIP path advertises, outputs + all gradients vs the oracle"""
    pc.check_model_vs_oracle(DEV, [(23, 31), (40, 17)], layers=3, seed=5, pair_seed=7, args_over=over, what=str(over),
                             l2=pc.GRAD_L2_SMALL, mx=pc.GRAD_MX_SMALL)


def test_many_pairs_split_head_backward():
    """more than 16 pairs: the keypoint head's weight-gradient kernel (k_head_u_bwd) runs one block per (head, group of 32
    segments) and its per-group partials go through the pass's fixed-order reduction"""
    pc.check_model_vs_oracle(DEV, [(9 + i % 5, 12 - i % 4) for i in range(19)], layers=2, seed=4, pair_seed=11,
                             what='19 pairs', l2=pc.GRAD_L2_SMALL, mx=pc.GRAD_MX_SMALL)


def test_pair_losses():
    pc.check_pair_losses(DEV)


def test_pocket_ot():
    pc.check_pocket_ot(DEV)


def test_edge_saved_state_is_bit_identical(monkeypatch):
    pc.check_edge_saved_state(DEV, monkeypatch, sizes=((33, 47), (7, 40)), layers=2)      # (the GPU test: four pairs, three layers)


def test_rigid_augment():
    pc.check_rigid_augment(DEV)


def test_composite_training_step():
    """model -> MSE + pocket OT + intersection (src/train.py:112-150) -> backward: loss and every parameter gradient vs the
    oracle's restatement of the same recipe"""
    pc.check_composite_training_step(DEV, sizes=((21, 37), (46, 18), (33, 40), (13, 25)), layers=3)
    pc.check_train_step_forms(DEV, sizes=((21, 37), (46, 18), (33, 40), (13, 25)), layers=2)
    pc.check_train_step_dropout(DEV, sizes=((21, 37), (26, 18)), layers=2)


def test_protein_graph_vs_reference_golden():
    pc.check_protein_graph(DEV)


@pytest.mark.parametrize('name', ['tiny', 'pair300'])
def test_protein_graph_more_reference_complexes(name):
    pc.check_protein_graph_case(DEV, name)


def test_real_structure_to_outputs_vs_reference():
    """1GL1: atoms -> graph kernels -> model kernels (all on the simulator) against the reference run end to end
    (tests/golden/case_F_real_1GL1.npz; the GPU suite also runs 2J7P and the two-complex batch)"""
    pc.check_real_structure_pipeline(DEV, 'F_real_1GL1')


def test_inference_postprocessing():
    pc.check_inference_postprocessing(DEV)


def test_fused_forward_launch_is_bit_identical():
    pc.check_fused_forward(DEV)


def test_gather_rides_in_the_attention_backward_launch():
    pc.check_gather_rides_in_attention_backward(DEV, sizes=((40, 35), (7, 70), (33, 16)))


@pytest.mark.parametrize('d', [64, 80])
def test_cross_attention_bf16(d):
    pc.check_attention_bf16(DEV, d)


@pytest.mark.parametrize('env', [dict(EQD_ATT_LB='0'), dict(EQD_ATT_SPLIT='0'), dict(EQD_ATT_SPLIT='1'),
                                 dict(EQD_ATT_LB_NB='2', EQD_ATT_SPLIT='0')], ids=str)
def test_cross_attention_bf16_kernel_forms(env, monkeypatch):
    """bf16 attention, d = 64: tiles held in LDS as bf16 (default; 16- and 32-row blocks, forward and backward) and the first
    version with fp32 tiles in LDS (EQD_ATT_LB=0) - same rounding points, same tolerance"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.check_attention_bf16(DEV, 64)
    pc.check_attention_bf16(DEV, 64, sizes=((300, 257), (129, 64)))


def test_results_do_not_depend_on_workspace_contents():
    pc.check_poisoned_workspaces(DEV, sizes=((36, 41), (7, 50), (20, 16)))      # (the GPU test runs larger pairs)


def test_attention_backward_ds_handoff_in_model():
    small = ((40, 35), (7, 70), (33, 16))      # (the GPU test runs larger pairs)
    pc.check_attention_ds_in_model(DEV, sizes=small)
    pc.check_attention_ds_in_model(DEV, bf16=True, sizes=small)


def test_linear_atb_bf16():
    pc.check_linear_atb_bf16(DEV)


def test_stack_backward_fp32_and_bf16():
    """backward of the layer stack from a fixed d(h_L, x_L): every layer parameter's gradient vs the oracle, plain"""
    pc.check_stack_backward(DEV, [(60, 75), (90, 48)], layers=4, seed=5, pair_seed=7, faithful=True, what='sim fp32')
    pc.check_stack_backward(DEV, [(60, 75), (90, 48)], layers=4, seed=5, pair_seed=7, faithful=True, bf16=True, what='sim bf16')


def test_head_backward_fp32_and_bf16():
    """the keypoint / Kabsch head alone, from the library's own last-layer state: outputs and d(h_L, x_L) + head parameter
    gradients vs the oracle's head, plain (eqd_model_head_backward)"""
    pc.check_head_backward(DEV, [(60, 75), (90, 48)], layers=3, seed=5, pair_seed=7, what='sim fp32')
    pc.check_head_backward(DEV, [(60, 75), (90, 48)], layers=3, seed=5, pair_seed=7, bf16=True, what='sim bf16')


def test_model_bf16_layer_states():
    pc.check_model_bf16_states(DEV, [(60, 75), (90, 48), (120, 100)], layers=4, seed=5, pair_seed=7, faithful=True, what='sim')


def test_inference_pipeline_end_to_end():
    pc.check_inference_pipeline(DEV)


def test_scalar_loss():
    pc.check_scalar_loss(DEV)


def test_coordinates_are_reread():
    pc.check_coords_reread(DEV)


def test_properties():
    pc.check_properties(DEV, sizes=((30, 41), (52, 27)), layers=2)


def test_cpu_tensor_rejected_by_real_library_contract():
    """The product refuses CPU tensors unless the explicitly loaded library is the simulator."""
    t = torch.zeros(3)
    _lib._is_sim = False
    try:
        with pytest.raises(_lib.EquidockHipError):
            _lib.require_device(t, 'x')
    finally:
        _lib._is_sim = True
