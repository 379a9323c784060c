"""GPU parity tests (-m gpu): the HIP path through the C ABI on a real MI355X against torch fp32
restatements, the golden vectors captured from the imported reference, and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from equidock_public_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.unload_for_testing()
    _lib.load_library()
    assert not _lib.is_simulator(), "GPU tests must run the real gfx950 library"
    return torch.device('cuda:0')


def test_linear(dev):
    from tests import parity_common as pc
    pc.check_linear(dev)


def test_linear_simple_form_is_bit_identical(dev, monkeypatch):
    """k_linear_simple (plain 64 x 64 projections, five workgroups per CU) against k_linear: same bits"""
    from tests import parity_common as pc
    pc.check_linear_simple_form(dev, monkeypatch)


def test_linear_simple80_form_is_bit_identical(dev, monkeypatch):
    """k_linear_simple80 (the first layer's 69-wide projection group, four workgroups per CU) against k_linear: same bits"""
    from tests import parity_common as pc
    pc.check_linear_simple80_form(dev, monkeypatch)


def test_training_step_is_bit_reproducible(dev):
    """run-to-run bits of a seeded training step (bf16 + dropout at DB5.5 batch size: the case that was not, round 6)"""
    from tests import parity_common as pc
    pc.check_run_to_run_bits(dev)


def test_atb(dev):
    from tests import parity_common as pc
    pc.check_atb(dev)


def test_edge_message_fwd_bwd(dev):
    from tests import parity_common as pc
    pc.check_edge(dev)


def test_edge_message_bf16(dev):
    from tests import parity_common as pc
    pc.check_edge_bf16(dev)


def test_edge_message_with_dropout_masks(dev):
    from tests import parity_common as pc
    pc.check_edge(dev, drop=True)
    pc.check_edge(dev, drop=True, bf16=True)


def test_dropout_training_through_the_kernels(dev):
    """dropout > 0 while training: HIP path vs the reference's recorded vectors and vs the torch-operator restatement"""
    from tests import parity_common as pc
    pc.check_dropout_training(dev)


def test_dropout_masks_drawn_by_the_library(dev):
    """hip_dropout_masks='library': eqd_dropout_draw vs its Philox restatement (bit-exact), then the model with those masks
    vs the torch-operator restatement handed the same masks"""
    from tests import parity_common as pc
    pc.check_dropout_library(dev)


@pytest.mark.parametrize('d', [64, 69, 80])
def test_cross_attention(dev, d):
    from tests import parity_common as pc
    pc.check_attention(dev, d)
    pc.check_attention(dev, d, sizes=((300, 257), (129, 64)))


@pytest.mark.parametrize('split', ['0', '1'])
def test_attention_half_blocks(dev, split, monkeypatch):
    """attention with one or two workgroups per 32-row work item (EQD_ATT_SPLIT forward, EQD_ATT_BWD_SPLIT backward)
    on the float4 paths, and a whole model without the splits (the defaults use them)"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_ATT_SPLIT', split)
    monkeypatch.setenv('EQD_ATT_BWD_SPLIT', split)
    pc.check_attention(dev, 64)
    pc.check_attention(dev, 80, sizes=((300, 257), (129, 64)))
    if split == '0':
        pc.check_model_case(dev, 'D_degraded3')


def test_kabsch(dev):
    from tests import parity_common as pc
    pc.check_kabsch(dev)


def test_keypoints_and_apply(dev):
    from tests import parity_common as pc
    pc.check_keypoints_and_apply(dev)


@pytest.mark.parametrize('K', [3, 16, 37, 64, 70])
def test_keypoint_head_counts(dev, K, monkeypatch):
    """num_att_heads other than 50: fewer heads than a lane group holds, whole 16-head blocks, a ragged last block, the
    largest count of the product backward (64) and one beyond it (the backward falls back to the one-head kernels)"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_KEYPOINT_MM', '1')
    pc.check_keypoints_and_apply(dev, K=K)


@pytest.mark.parametrize('form', ['first', 'mm', 'mm_chunks', 'mm_long'])
def test_keypoint_kernel_forms(dev, form, monkeypatch):
    """both sets of keypoint-pooling kernels (one workgroup per (segment, head); the matrix-product forms with one / several
    row chunks per segment) against torch, the golden vectors and the oracle's head backward"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_KEYPOINT_MM', '0' if form == 'first' else '1')
    if form == 'mm_long':      # one long segment: the forward's 16-wave workgroups, a backward of 8 chunks (some of them empty)
        pc.check_keypoints_and_apply(dev, sizes=((1030, 40),))
        return
    if form == 'mm_chunks':
        monkeypatch.setenv('EQD_KEYPOINT_NC', '2')
        pc.check_keypoints_and_apply(dev, sizes=((150, 130), (140, 161)))
        pc.check_head_backward(dev, [(300, 280), (290, 310)], layers=3, what=f'keypoint kernels: {form}')
    else:
        pc.check_keypoints_and_apply(dev)
        pc.check_head_backward(dev, [(41, 57), (66, 38)], layers=3, what=f'keypoint kernels: {form}')
    pc.check_model_case(dev, 'D_degraded3')
    if form == 'first':      # 18 pairs = 36 segments: the key / query maps' backward in two segment groups (partial sums); the
                             # product kernels' groups: test_many_pairs_split_head_backward
        pc.check_keypoints_and_apply(dev, sizes=tuple((12 + i, 30 - i) for i in range(18)))
        pc.check_head_backward(dev, [(12 + i, 30 - i) for i in range(18)], layers=2, what=f'keypoint kernels, 18 pairs: {form}')
    names = pc.launch_names_of_a_step(dev, 'D_degraded3')
    assert ('k_keypoint_bwd' in names) == (form != 'first') and ('k_keypoint_bwd_a' in names) == (form == 'first'), sorted(set(names))


@pytest.mark.parametrize('name', ['A_b1_shared5', 'B_b3_dips8', 'C_b2_200', 'D_degraded3', 'E_svd_guard',
                                  'F_real_1GL1', 'F_real_2J7P', 'F_real_batch2'])
def test_model_vs_golden(dev, name):
    """F_real_*: the real reference model on the graphs the reference's own builder made from real DB5.5 structures
    (oracle/make_golden_real.py)"""
    from tests import parity_common as pc
    pc.check_model_case(dev, name)


@pytest.mark.parametrize('name', ['F_real_1GL1', 'F_real_2J7P', 'F_real_batch2'])
def test_real_structure_to_outputs_vs_reference(dev, name):
    """atoms -> HIP graph kernels -> HIP model against the reference run end to end on the same structure"""
    from tests import parity_common as pc
    REPORT.append(pc.check_real_structure_pipeline(dev, name))


def test_model_bf16_on_real_structures(dev):
    """bf16 mode on the graphs the reference's own builder made from 1GL1 + 2J7P (tests/golden/case_F_real_batch2.npz), 8 layers:
    last-layer state and the five outputs within 2 x the bf16 oracle's own distance to its 1e-6-perturbed copies (the measured
    bound of check_model_bf16_states), on coordinates that are not centred"""
    from tests import parity_common as pc
    from tests.util import load_case, pairs_from_raw
    z, meta, args, raw = load_case('F_real_batch2')
    pc.check_model_bf16_states(dev, None, layers=8, seed=3, what='real structures 1GL1 + 2J7P, bf16', report=REPORT,
                               pairs=pairs_from_raw(raw))


def test_real_ragged_batch_vs_oracle(dev):
    """1DE4 (1 270 + 40 residues) beside 2J7P (259 + 286): real graphs from the HIP graph kernels, model vs the oracle"""
    from tests import parity_common as pc
    pc.check_real_ragged_batch_vs_oracle(dev, report=REPORT)


@pytest.mark.parametrize('name', ['D_degraded3', 'B_b3_dips8'])
def test_model_two_row_tiles(dev, name, monkeypatch):
    """the 32-rows-per-workgroup variants of the row kernels (EQD_ROW_TILES=2) on small golden cases"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_ROW_TILES', '2')
    monkeypatch.setenv('EQD_ROWWAVE', '0')
    pc.check_linear(dev)
    pc.check_model_case(dev, name)


@pytest.mark.parametrize('rowwave', ['0', '1', '2'])
def test_row_kernels_both_forms(dev, rowwave, monkeypatch):
    """k_rowwave (one wave per 16-row tile, chain in registers: every node-level chain of the 64-wide layers) and
    k_rowchain / k_linear (four waves per tile: layer 0, and everything under EQD_ROWWAVE=0) against the golden vectors
    and the oracle; the switch must really select the kernel."""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_ROWWAVE', rowwave)
    pc.check_linear(dev)
    pc.check_linear_atb_bf16(dev)
    for name in ('D_degraded3', 'B_b3_dips8'):
        pc.check_model_case(dev, name)
    pc.check_model_bf16(dev, 'D_degraded3')
    pc.check_dropout_library(dev)           # the dropout factors in each form's epilogue and LayerNorm backward
    names = pc.launch_names_of_a_step(dev, 'D_degraded3')
    assert ('k_rowwave' in names) == (rowwave == '1') and ('k_rowres' in names) == (rowwave == '2'), sorted(set(names))
    assert 'k_rowchain' in names


@pytest.mark.parametrize('tps', ['3', '11', '16'])
def test_rowres_tiles_per_workgroup(dev, tps, monkeypatch):
    """k_rowres with several tiles per workgroup (one and two tile slots per wave, idle waves, ragged last workgroup)"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_ROWWAVE', '2')
    monkeypatch.setenv('EQD_ROWRES_TPS', tps)
    pc.check_linear(dev)
    pc.check_linear_atb_bf16(dev)
    for name in ('D_degraded3', 'B_b3_dips8'):
        pc.check_model_case(dev, name)
    pc.check_model_bf16(dev, 'D_degraded3')


@pytest.mark.parametrize('name', ['D_degraded3', 'A_b1_shared5'])
def test_model_bf16_mode(dev, name):
    from tests import parity_common as pc
    pc.check_model_bf16(dev, name)


def test_model_forked_attention_stream(dev, monkeypatch):
    """EQD_FORK=1: the forward's attention kernel on the library's auxiliary stream (off by default)"""
    from tests import parity_common as pc
    monkeypatch.setenv('EQD_FORK', '1')
    pc.check_model_case(dev, 'D_degraded3')


def test_first_layer_on_rowres80(dev):
    """the 69-wide first layer's forward jobs (bf16 mode) on k_rowres80 and on the four-wave kernels vs torch with the mode's rounding points"""
    from tests import parity_common as pc
    pc.check_first_layer_rowres80(dev, rows=333)
    pc.check_first_layer_rowres80(dev, rows=5000)


def test_rowres80_dropout(dev):
    """training with dropout, bf16 mode: the first layer on k_rowres80 vs on the four-wave kernels, same library-drawn masks"""
    from tests import parity_common as pc
    pc.check_rowres80_dropout(dev)
    pc.check_rowres80_dropout(dev, sizes=((400, 380), (350, 410), (90, 120)))


def test_bf16_storage_operators(dev):
    """EqdLinJob.Yb (bf16 copy == RNE of the fp32 output, every epilogue) and EqdAtbJob.y_bf16 (same bits as the fp32 Y)"""
    from tests import parity_common as pc
    pc.check_bf16_storage_ops(dev)


def test_bf16_storage_of_the_saved_state(dev, monkeypatch):
    from tests import parity_common as pc
    pc.check_bf16_storage_model(dev, monkeypatch)


def test_lane_exchanges(dev):
    """DPP / v_permlane*_swap helpers vs __shfl_xor, bit for bit, inside one kernel"""
    from tests import parity_common as pc
    pc.check_lane_exchanges(dev)


def test_node_update_operator(dev):
    """eqd_node_update_fwd / _bwd vs torch autograd of node_mlp + skip (64-wide, 69-wide with 80-float cross rows, no cross);
    at 301 rows (four-wave kernels) and at 40 000 rows (k_rowres for the 64-wide chains)"""
    from tests import parity_common as pc
    pc.check_node_update(dev, rows=301)
    pc.check_node_update(dev, rows=40000)


def test_standalone_layer_through_the_library(dev):
    """IEGMN_Layer.forward on its own: edge messages + cross attention in the HIP library vs the torch-operator restatement"""
    from tests import parity_common as pc
    pc.check_standalone_layer(dev)


def test_flat_grads(dev):
    from tests import parity_common as pc
    pc.check_flat_grads_equal_autograd(dev)


def test_model_vs_oracle_ragged(dev):
    from tests import parity_common as pc
    # 1-node proteins run the Kabsch guard loop (A = 0): outputs are compared, gradients only checked to be finite
    # (the closed-form SVD backward divides by singular-value gaps of the random diagonal there)
    pc.check_model_vs_oracle_ragged(dev, check_grads=False)
    pc.check_model_vs_oracle_ragged(dev, sizes=((4, 4), (17, 5), (33, 64)))
    # seeds 8, 9 and 11 each have a LeakyReLU pre-activation within fp32 rounding of 0 that takes the other slope on the
    # GPU than in the oracle's own evaluation (x100 on that derivative at the reference's slope 0.01).  Since round 3 the
    # oracle is evaluated with the library's own decisions (parity_common.oracle_given), so these runs are held to the
    # same plain tolerance as everything else (until round 2: 5e-3 / 2e-2 against a hull); one more run with slope 0.5.
    for seed in (8, 9, 10, 11):
        pc.check_model_vs_oracle_ragged(dev, sizes=((129, 257), (300, 31), (64, 64), (95, 200)), layers=3, seed=seed)
    pc.check_model_vs_oracle_ragged(dev, sizes=((129, 257), (300, 31), (64, 64), (95, 200)), layers=3, seed=8, slope=0.5)


@pytest.mark.parametrize('over', [dict(cross_msgs=False), dict(use_dist_in_layers=False),
                                  dict(use_edge_features_in_gmn=False), dict(use_mean_node_features=False),
                                  dict(x_connection_init=0.25), dict(num_att_heads=13),
                                  dict(cross_msgs=False, use_mean_node_features=False, x_connection_init=0.25,
                                       shared_layers=True)],
                         ids=lambda o: '+'.join(f'{k}={v}' for k, v in o.items()))
def test_option_toggles_vs_oracle(dev, over):
    """every `args` switch the HIP path advertises, outputs + all gradients vs the oracle"""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(23, 31), (40, 17), (130, 77)], layers=3, seed=5, pair_seed=7, args_over=over,
                             what=str(over), l2=pc.GRAD_L2_SMALL, mx=pc.GRAD_MX_SMALL)


def test_many_pairs_split_head_backward(dev):
    """more than 16 pairs: k_head_u_bwd on (head, segment group) blocks + fixed-order reduction of the group partials"""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(9 + i % 5, 12 - i % 4) for i in range(19)], layers=2, seed=4, pair_seed=11,
                             what='19 pairs', l2=pc.GRAD_L2_SMALL, mx=pc.GRAD_MX_SMALL)
    pc.check_model_vs_oracle(dev, [(40 + i % 7, 35 + i % 3) for i in range(40)], layers=3, seed=5, pair_seed=12,
                             what='40 pairs', l2=pc.GRAD_L2_SMALL, mx=pc.GRAD_MX_SMALL)


def test_pair_losses(dev):
    from tests import parity_common as pc
    pc.check_pair_losses(dev)


def test_pocket_ot(dev):
    from tests import parity_common as pc
    pc.check_pocket_ot(dev)


def test_composite_training_step(dev):
    """the reference's training step (model -> MSE + pocket OT + intersection -> backward) through the library: loss and every
    parameter gradient vs the oracle (src/train.py:98-154)"""
    from tests import parity_common as pc
    pc.check_composite_training_step(dev, report=REPORT)


@pytest.mark.parametrize('bf16', [False, True])
def test_train_step_graphs_around_one_host_join(dev, bf16):
    """train_step.TrainStep: forward / pair terms / backward as three hipGraphs around the exact transport solve on the host -
    same loss and flat gradient as the autograd form"""
    from tests import parity_common as pc
    pc.check_train_step_forms(dev, bf16=bf16)
    pc.check_train_step_forms(dev, sizes=[(200, 200)] * 8, layers=8, bf16=bf16)


def test_train_step_with_dropout_masks_per_replay(dev):
    """TrainStep with dropout 0.25: fresh library-drawn masks in every replay of the captured graphs"""
    from tests import parity_common as pc
    pc.check_train_step_dropout(dev)


def test_edge_saved_state_is_bit_identical(dev, monkeypatch):
    """the per-edge state a training forward saves for the edge backward (round 6) against the recompute: same bits"""
    from tests import parity_common as pc
    pc.check_edge_saved_state(dev, monkeypatch)


def test_rigid_augment(dev):
    from tests import parity_common as pc
    pc.check_rigid_augment(dev)


def test_protein_graph_vs_reference_golden(dev):
    from tests import parity_common as pc
    pc.check_protein_graph(dev)


@pytest.mark.parametrize('name', ['tiny', 'pair300', 'big'])
def test_protein_graph_more_reference_complexes(dev, name):
    """graph construction on three more DB5.5 complexes (a 1 270-residue protein, a DIPS-sized pair, fewer residues than
    max_neighbor) against the reference's own function: bit-exact int32 endpoints"""
    from tests import parity_common as pc
    pc.check_protein_graph_case(dev, name)


def test_inference_postprocessing(dev):
    from tests import parity_common as pc
    pc.check_inference_postprocessing(dev)


def test_fused_forward_launch_is_bit_identical(dev):
    from tests import parity_common as pc
    pc.check_fused_forward(dev)


def test_gather_rides_in_the_attention_backward_launch(dev):
    """k_attn_bwd_gather (attention backward + node gather + reductions in one launch) vs the separate launches: bit-identical"""
    from tests import parity_common as pc
    pc.check_gather_rides_in_attention_backward(dev)


@pytest.mark.parametrize('d', [64, 80])
def test_cross_attention_bf16(dev, d):
    from tests import parity_common as pc
    pc.check_attention_bf16(dev, d)
    pc.check_attention_bf16(dev, d, sizes=((300, 257), (129, 64)))


@pytest.mark.parametrize('env', [dict(EQD_ATT_LB='0'), dict(EQD_ATT_SPLIT='0'), dict(EQD_ATT_SPLIT='1'),
                                 dict(EQD_ATT_LB_NB='2', EQD_ATT_SPLIT='0')], ids=str)
def test_cross_attention_bf16_kernel_forms(dev, env, monkeypatch):
    """bf16 attention, d = 64: tiles held in LDS as bf16 (default; 16- and 32-row blocks, forward and backward) and the first
    version with fp32 tiles in LDS (EQD_ATT_LB=0) - same rounding points, same tolerance"""
    from tests import parity_common as pc
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pc.check_attention_bf16(dev, 64)
    pc.check_attention_bf16(dev, 64, sizes=((300, 257), (129, 64)))


def test_linear_atb_bf16(dev):
    from tests import parity_common as pc
    pc.check_linear_atb_bf16(dev)


def test_inference_pipeline_end_to_end(dev):
    from tests import parity_common as pc
    pc.check_inference_pipeline(dev)


def test_scalar_loss(dev):
    from tests import parity_common as pc
    pc.check_scalar_loss(dev)


def test_coordinates_are_reread(dev):
    from tests import parity_common as pc
    pc.check_coords_reread(dev)


def test_properties_small(dev):
    from tests import parity_common as pc
    pc.check_properties(dev)


def test_properties_baseline_sizes(dev):
    """Size-independent properties at BASELINE.json's full sizes (config B graphs, 8 layers)."""
    from tests import parity_common as pc
    pc.check_properties(dev, sizes=((200, 200), (200, 200), (200, 200), (200, 200)), layers=8)


REPORT = []      # one line per BASELINE.json workload, printed at the end of the module (pytest -s) and kept in
                 # gpurun_out/parity_report.txt when that directory exists


@pytest.fixture(scope='module', autouse=True)
def _parity_report():
    yield
    if REPORT:
        import os
        text = '\n'.join(REPORT) + '\n'
        print('\n' + text)
        if os.path.isdir('gpurun_out'):
            with open('gpurun_out/parity_report.txt', 'a') as f:
                f.write(text)


def test_model_vs_oracle_config_b(dev):
    """BASELINE.json configs[1] at its real workload: 8 x (200, 200), 8 layers, fp32 - outputs and every gradient vs
    the oracle in its faithful mode (the reference's op sequence, dense batch-wide mask)."""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(200, 200)] * 8, layers=8, seed=3, pair_seed=33, faithful=True, what='config B fp32',
                             report=REPORT)


def test_model_vs_oracle_config_c_fp32(dev):
    """BASELINE.json configs[2] shapes at the real workload, fp32 arithmetic: 64 x (300, 300), 8 layers.  The dense mask
    of the faithful oracle would be 19 200^2 per temporary, so the oracle runs its block-diagonal mode (equal to the
    reference to 0.0 on every golden case, tests/test_oracle_golden.py)."""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(300, 300)] * 64, layers=8, seed=3, pair_seed=34, faithful=False, what='config C fp32',
                             report=REPORT)


def test_model_vs_oracle_config_c_bf16(dev):
    """BASELINE.json configs[2] as stated (bf16): 64 x (300, 300), 8 layers, hip_storage_dtype='bf16' - the state after the
    last IEGMN layer against the oracle evaluated with the bf16 mode's rounding points (see check_model_bf16_states for why
    the comparison is made there and not on the final outputs)."""
    from tests import parity_common as pc
    pc.check_model_bf16_states(dev, [(300, 300)] * 64, layers=8, seed=3, pair_seed=34, what='config C bf16', report=REPORT)


def test_results_do_not_depend_on_workspace_contents(dev):
    from tests import parity_common as pc
    pc.check_poisoned_workspaces(dev)


def test_attention_backward_ds_handoff_in_model(dev):
    """the dS hand-off form of the attention backward (large batches) against the recompute form, whole model"""
    from tests import parity_common as pc
    pc.check_attention_ds_in_model(dev)
    pc.check_attention_ds_in_model(dev, bf16=True)


def test_stack_backward_config_c_bf16(dev):
    """BASELINE.json configs[2] as stated (bf16, 64 x (300, 300), 8 layers, ROT scale 40 like the bench): the backward of the
    layer stack from a fixed gradient w.r.t. the last layer's state - every layer parameter's gradient against the oracle
    with the bf16 mode's rounding points and the library's LeakyReLU decisions (parity_common.check_stack_backward)."""
    from tests import parity_common as pc
    pc.check_stack_backward(dev, [(300, 300)] * 64, layers=8, seed=3, pair_seed=34, bf16=True, what='config C bf16', report=REPORT)


def test_head_backward_config_c_bf16(dev):
    """BASELINE.json configs[2] (bf16, 64 x (300, 300), 8 layers, ROT scale 40): the keypoint / Kabsch head differentiated on
    its own from the library's (h_L, x_L) - d(h_L, x_L) and the head parameters' gradients against the oracle's head,
    plainly.  With the stack test above it covers the whole bf16 model."""
    from tests import parity_common as pc
    pc.check_head_backward(dev, [(300, 300)] * 64, layers=8, seed=3, pair_seed=34, bf16=True, what='config C bf16', report=REPORT)


def test_head_backward_config_b_fp32(dev):
    from tests import parity_common as pc
    pc.check_head_backward(dev, [(200, 200)] * 8, layers=8, seed=3, pair_seed=33, what='config B fp32', report=REPORT)


def test_stack_backward_config_b_fp32(dev):
    from tests import parity_common as pc
    pc.check_stack_backward(dev, [(200, 200)] * 8, layers=8, seed=3, pair_seed=33, faithful=True, what='config B fp32',
                            report=REPORT)


def test_model_vs_oracle_config_e(dev):
    """BASELINE.json configs[4] at its real workload: 4 x (2000, 2000), 8 layers, fp32, vs the oracle (block-diagonal
    mode: four 2000 x 2000 attention blocks per direction on the host)."""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(2000, 2000)] * 4, layers=8, seed=3, pair_seed=35, faithful=False, what='config E fp32',
                             report=REPORT)


def test_model_vs_oracle_config_a(dev):
    """BASELINE.json configs[0]: one (200, 200) pair, 5-layer shared IEGMN."""
    from tests import parity_common as pc
    pc.check_model_vs_oracle(dev, [(200, 200)], layers=5, seed=3, pair_seed=36, faithful=True, what='config A fp32',
                             args_over=dict(shared_layers=True, skip_weight_h=0.5), report=REPORT)


def test_model_vs_oracle_workload_r(dev):
    """`bench.py --workload R` (SURVEY.md section 8d "realistic sizes": 64 ragged pairs, ligand 29..1500 / receptor 40..2130
    residues): the first 16 of those pairs (56..856 residues per protein) at 8 layers against the oracle."""
    import bench
    from equidock_public_amd import synthetic
    from tests import parity_common as pc
    sizes = synthetic.realistic_sizes(64, bench.R_SIZE_SEED)[:16]
    pc.check_model_vs_oracle(dev, sizes, layers=8, seed=3, pair_seed=37, faithful=False, what='workload R (16 of 64 pairs)',
                             report=REPORT)


def test_bench_workload_d_on_rccl_world_of_one(dev):
    """`bench.py --workload D` (BASELINE.json configs[3]: 64 x (300, 300) per GPU, bf16, data parallel) launched the way the
    driver launches N > 1 - torch.distributed.run, backend nccl (= RCCL) - with one rank: the flat-gradient all-reduce runs
    on RCCL every step (inside the replayed hipGraph when RCCL accepts the capture) and its cost is in the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(root, 'bench.py'), '--gpus', '1', '--workload', 'D', '--steps', '3',
           '--warmup', '2', '--no-cpu-baseline', '--no-roofline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['rccl_ranks'] == 1 and out['n_gpus'] == 1 and out['scaling'] == 'weak'
    assert out['allreduce_us_per_step'] is not None and out['allreduce_us_per_step'] > 0
    assert out['allreduce_bytes'] > 3_000_000 and out['value'] > 0
    assert 'bf16' in out['dtype'] and out['config']['pairs_per_gpu'] == 64
    REPORT.append(f"bench.py --workload D under torch.distributed.run (nccl, 1 rank): {out['value']} pairs/s, "
                  f"all-reduce of {out['allreduce_bytes']} B: {out['allreduce_us_per_step']} us per step, "
                  f"in the replayed hipGraph: {out['allreduce_in_graph']}")


def test_bench_gpus2_plain_command_two_ranks_on_one_gpu(dev):
    """`python bench.py --gpus 2 --steps 2` with NO launcher: bench.py re-executes itself under torch.distributed.run.  On
    this one-GPU box both ranks share cuda:0 over gloo (EQD_BENCH_ONE_DEVICE=1, EQD_BENCH_BACKEND=gloo: a rehearsal of the
    N > 1 control flow - rendezvous, barriers, max over ranks, the flat-gradient all-reduce every step, one line from rank
    0 - not a measurement).  The secondary workload of a distributed default run (D) rides along."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', EQD_BENCH_ONE_DEVICE='1', EQD_BENCH_BACKEND='gloo')
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--no-roofline', '--no-secondary']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['config']['parallelism'] == 'dp2' and out['config']['pairs_per_gpu'] == 8
    assert out['allreduce_us_per_step'] is not None
    REPORT.append(f"python bench.py --gpus 2 (no launcher; self-relaunch, 2 ranks on one GPU over gloo): rc 0, one line, "
                  f"{out['value']} pairs/s (rehearsal, not a measurement)")


def test_big_batch_equals_small_batches(dev):
    """Config C regime (> 16 384 nodes: several super-tiles per workgroup in the edge backward, capped AtB parts):
    pairs are independent, so outputs must equal those of the same pairs run in batches of 4 (different kernel decompositions), and the gradient of the summed loss must equal the sum."""
    from equidock_public_amd import graph as G, synthetic
    from oracle import iegmn_port as port
    from tests import parity_common as pc
    args = port.default_args(iegmn_n_lays=3, skip_weight_h=0.75)
    sd = port.init_state_dict(args, seed=9)
    net = pc.build_model(args, sd, dev)
    sizes = [(180 + 7 * (i % 5), 200 - 3 * (i % 7)) for i in range(46)]
    pairs = synthetic.make_pairs(sizes, 77)
    assert sum(a + b for a, b in sizes) > 16384

    def run(ps):
        g = G.batch_pairs(ps).to(dev)
        for p in net.parameters():
            p.grad = None
        lig, Yl, Yr, T, b = net.forward_batched(g)
        (lig.square().sum() * 1e-3 + Yl.square().sum() * 1e-3 + Yr.square().sum() * 1e-3).backward()
        return [t.detach().cpu() for t in (lig, Yl, Yr, T, b)], {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
    big, gbig = run(pairs)
    parts, gsum = [], None
    for i in range(0, len(pairs), 4):
        o, gr = run(pairs[i:i + 4])
        parts.append(o)
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
    for j, nm in enumerate(('lig', 'Yl', 'Yr', 'T', 'b')):
        pc.close(big[j], torch.cat([p[j] for p in parts], 0), tol=1e-4, what=f'big batch vs small batches: {nm}')
    for k in gbig:
        pc.grad_close(gbig[k], gsum[k], what=f'big batch vs small batches: grad {k}')


def test_pack_on_host_then_move(dev):
    """collate-in-a-worker pattern: batch + pack on the host, one move to the GPU, same outputs as packing on the GPU"""
    from equidock_public_amd import graph as G, synthetic
    from oracle import iegmn_port as port
    from tests import parity_common as pc
    args = port.default_args(iegmn_n_lays=2, skip_weight_h=0.5)
    net = pc.build_model(args, port.init_state_dict(args, seed=5), dev)
    pairs = synthetic.make_pairs([(50, 61), (33, 47), (80, 20)], 5)
    g_host = G.batch_pairs(pairs)
    g_host.pack()                       # CPU layout (what a DataLoader worker would produce)
    g_moved = g_host.to(dev)
    assert g_moved.pack().he_bf16.is_cuda and g_moved.pack().he_bf16.data_ptr() % 16 == 0
    with torch.no_grad():
        a = net.forward_batched(g_moved)
        b = net.forward_batched(G.batch_pairs(pairs).to(dev))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # through a DataLoader: collate_fn builds and packs the batch, pin_memory=True pins its per-dtype buffers
    # (PairGraph.pin_memory), .to() is then asynchronous copies from page-locked memory
    from torch.utils.data import DataLoader

    def collate(items):
        gb = G.batch_pairs(items)
        gb.pack()
        return gb
    for workers in (0, 2):
        dl = DataLoader(pairs + pairs, batch_size=3, shuffle=False, collate_fn=collate, pin_memory=True, num_workers=workers)
        batches = list(dl)
        assert len(batches) == 2 and batches[0].pack().he.is_pinned()
        with torch.no_grad():
            c = net.forward_batched(batches[1].to(dev))
        for x, y in zip(a, c):
            assert torch.equal(x, y)


def test_large_complex_runs(dev):
    """Stress shape (one 2000 + 2000 residue pair, 2 layers): finite outputs, valid rotation."""
    from equidock_public_amd import graph as G, synthetic
    from oracle import iegmn_port as port
    from tests import parity_common as pc
    args = port.default_args(iegmn_n_lays=2, skip_weight_h=0.5)
    net = pc.build_model(args, port.init_state_dict(args, seed=4), dev)
    g = G.batch_pairs(synthetic.make_pairs([(2000, 2000)], 44)).to(dev)
    lig, Yl, Yr, T, b = net.forward_batched(g)
    (lig.square().mean() + Yl.square().mean()).backward()
    torch.cuda.synchronize()
    assert torch.isfinite(lig).all() and torch.isfinite(T).all()
    t = T[0].detach().cpu()
    np.testing.assert_allclose((t @ t.t()).numpy(), np.eye(3), atol=1e-5)
    for p in net.parameters():
        assert torch.isfinite(p.grad).all()


def test_smoke_entry(dev):
    import __graft_entry__ as ge
    ge.smoke()


def test_flat_grad_allreduce_on_rccl_world_of_one(dev):
    """The data-parallel exchange (parallel.FlatGradAllReduce: ONE all-reduce of the flat gradient buffer) on backend
    'nccl' (= RCCL) in a world of one: the collective runs on the GPU buffer and leaves the gradient unchanged."""
    import torch.distributed as dist
    from equidock_public_amd import graph as G, parallel, synthetic
    from oracle import iegmn_port as port
    from tests import parity_common as pc
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1, device_id=dev)
    try:
        args = port.default_args(iegmn_n_lays=2, skip_weight_h=0.5)
        net = pc.build_model(args, port.init_state_dict(args, seed=5), dev)
        red = parallel.FlatGradAllReduce(net)
        g = G.batch_pairs(synthetic.make_pairs([(50, 61), (33, 47)], 5)).to(dev)
        red.zero()
        port.scalar_loss(net(g, epoch=0)).backward()
        before = red.flat.clone()
        red.reduce(force=True)
        torch.cuda.synchronize()
        assert float(before.abs().sum()) > 0 and torch.equal(before, red.flat)
        parallel.broadcast_parameters(net)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['swish_bn_gn_eval', 'fine_tune'])
def test_non_published_options_on_gpu(dev, name):
    """The torch-operator path of the reference's non-published options (tests/test_torch_path.py pins it on the CPU) on the
    GPU: same vectors from the real reference module; the fine-tune model runs its first stage in the HIP library and hands
    the docked ligand, with its gradient, to the torch-path second stage."""
    import json
    import os
    from equidock_public_amd import config, graph as G, model as M
    from oracle import iegmn_port as port
    from tests.util import cat_out, pairs_from_raw
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'variants.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    v = meta['variants'][name]
    args = dict(v['args'], device=dev)
    net = M.Rigid_Body_Docking_Net(args).to(dev)
    net.load_state_dict(config.seeded_state_dict(args, meta['init_seed'], meta['rot_scale']))
    net.train(v['train'])
    raw = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in_')}
    raw['lig_counts'], raw['rec_counts'] = [int(c) for c in z['in_lig_counts']], [int(c) for c in z['in_rec_counts']]
    g = G.batch_pairs(pairs_from_raw(raw)).to(dev)
    outs = net(g, epoch=0)
    for nm, lst in zip(('lig', 'Yl', 'Yr', 'T', 'b'), outs):
        got, ref = cat_out(lst).detach().cpu(), torch.from_numpy(z[f'{name}_{nm}'])
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), (name, nm)
    loss = port.scalar_loss(outs)
    loss.backward()
    assert abs(float(loss.detach()) - float(z[f'{name}_loss'])) <= 1e-4 * abs(float(z[f'{name}_loss']))
    # (biases in front of a BatchNorm / GraphNorm: mathematically zero gradient, computed value = rounding noise of the
    # device's summation order - hence the floor relative to the largest gradient norm, as in tests/test_torch_path.py)
    floor = 1e-6 * max(v['grad_norms'].values()) + 1e-5
    for k, p in net.named_parameters():
        ref = v['grad_norms'][k]
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        assert abs(got - ref) <= 2e-3 * ref + floor, f'{name}: gradient norm of {k}: {got} vs {ref}'
