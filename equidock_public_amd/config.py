"""Configuration helpers of the drop-in: the `args` dictionary the reference's modules consume and seeded
PyTorch-default weights.

The reference builds `args` with a module-level argparse (src/utils/args.py:15-119; importing it parses sys.argv) and
gets its initial weights from the default initialisers of the torch modules it constructs
(src/model/rigid_docking_model.py:82-175, 360-440 - `reset_parameters` is never called, :175).  `published_args`
returns the published configuration family (src/utils/args.py:227-280, checkpoints src/inference_rigid.py:90,93) as a
plain dict with exactly the keys the three modules read; `seeded_state_dict` constructs the drop-in's parameter
containers on the host under a torch seed - same modules, same construction order, hence the reference's own initial
weights for that seed (tests/test_abi_and_graph.py pins this against the oracle's independent restatement).
No arithmetic of the hot path lives here.
"""
import torch


def published_args(**over):
    a = dict(input_edge_feats_dim=27, dropout=0.0, nonlin='lkyrelu', cross_msgs=True, layer_norm='LN',
             layer_norm_coors='0', final_h_layer_norm='0', use_dist_in_layers=True, skip_weight_h=0.75,
             x_connection_init=0.0, leakyrelu_neg_slope=0.01, debug=False, device=torch.device('cpu'),
             graph_nodes='residues', rot_model='kb_att', noise_decay_rate=0.0, noise_initial=0.0,
             use_edge_features_in_gmn=True, use_mean_node_features=True, residue_emb_dim=64,
             iegmn_lay_hid_dim=64, shared_layers=False, num_att_heads=50, iegmn_n_lays=8, fine_tune=False)
    a.update(over)
    return a


def seeded_state_dict(args, seed, rot_scale=40.0):
    """state_dict of a freshly constructed Rigid_Body_Docking_Net under torch seed `seed`; the keypoint key/query
    projections are scaled by `rot_scale` (with the default initialisation all keypoints nearly coincide and the SVD
    guard of rigid_docking_model.py:574-580 fires on every pair).  The global RNG state is
    restored."""
    from . import model
    state = torch.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = model.Rigid_Body_Docking_Net(dict(args, device=torch.device('cpu')))
    finally:
        torch.set_rng_state(state)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k in sd:        # both stages when fine_tune is on
        if k.endswith('att_mlp_key_ROT.0.weight') or k.endswith('att_mlp_query_ROT.0.weight'):
            sd[k] = sd[k] * rot_scale
    return sd
