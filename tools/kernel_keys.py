"""Kernel classification shared by the rocprofv3 summaries: the full-batch launch of each hot kernel of each workload."""
# key -> (substring of the kernel symbol, robot tag, minimal grid size (blocks * 64 lanes) of a full-batch launch)
KEYS = {
    "deriv_body": ("10deriv_body", "DimsILi13ELi4EEE", 4096 * 51 * 64),
    "riccati_kino_body": ("riccati_kino_body", "", 4096 * 64),
    "forward_kino_body": ("forward_kino_body", "", 4096 * 64),
    "trial_body": ("10trial_body", "", 4096 * 51 * 64),
    "lane_tree_body": ("lane_tree_body", "ELb1ELb0E", 51 * 64 * 64),     # derivative pass: lane-per-problem tree pass, stream hand-over (64 problems per block)
    "lane_tree_ls_body": ("lane_tree_body", "ELb0ELb0E", 51 * 64 * 64),  # line search: the same kernel in evaluation mode (heads into tiles)
    "trial_rows_body": ("trial_rows_body", "", 4096 * 51 * 64),    # line search: rows of the candidate (block per problem)
    "deriv2_body": ("deriv2_body", "", 4096 * 51 * 64),            # SMPC_LANE_DERIV=1 only
    "apply_body": ("apply_body", "DimsILi13ELi4EEE", 4096 * 64),
    "cent_step_body": ("cent_step_body", "", 4096 * 64),           # SMPC_CENT_FUSED=1 only (round 5: the pipeline below)
    "cent_pre_body": ("cent_pre_body", "", 4096 * 64),
    "cent_bwd_body": ("cent_bwd_body", "", 4096 * 64),
    "cent_fwd_body": ("cent_fwd_body", "", 4096 * 64),
    "cent_ls_body": ("cent_ls_", "", 4096 * 64),                   # cent_ls_poly_body (horizons <= 63) / cent_ls_body
    "fdyn_deriv_body_go2": ("fdyn_deriv_body", "FullDimsILi13E", 4096 * 51 * 64),
    "fdyn_trial_body_go2": ("fdyn_trial_body", "FullDimsILi13E", 4096 * 51 * 64),
    "riccati_dense_body_go2": ("riccati_dense_body", "FullDimsILi13E", 4096 * 64),
    "forward_full_body_go2": ("forward_full_body", "FullDimsILi13E", 4096 * 64),
    # Talos full dynamics = FullDims<23,2,6,0,0,0>; the kinodynamics OCP with 6-D feet runs the same kernels as FullDims<23,2,6,0,0,1>
    # (round 6: the biped's derivative kernel runs a persistent grid of compute units x 4 blocks; its full-batch launches are the TAG = 0 ones)
    "fdyn_deriv_body_talos": ("fdyn_deriv_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi0E", 1024 * 64),
    "fdyn_trial_body_talos": ("fdyn_trial_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi0E", 1024 * 101 * 64),
    "riccati_dense_body_talos": ("riccati_dense_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi0E", 1024 * 64),
    "forward_full_body_talos": ("forward_full_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi0E", 1024 * 64),
    "fdyn_deriv_body_taloskino": ("fdyn_deriv_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi1E", 1024 * 64),
    "fdyn_trial_body_taloskino": ("fdyn_trial_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi1E", 1024 * 101 * 64),
    "riccati_dense_body_taloskino": ("riccati_dense_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi1E", 1024 * 64),
    "forward_full_body_taloskino": ("forward_full_body", "FullDimsILi23ELi2ELi6ELi0ELi0ELi1E", 1024 * 64),
    # centroidal OCP with 6-D feet: (instance x stage) kernels around the dense sweep
    "cent6_deriv_body": ("cent6_deriv_body", "", 1024 * 101 * 64),
    "riccati_dense_body_cent6": ("riccati_dense_body", "Cent6DimsILi2E", 1024 * 64),
    "cent6_forward_body": ("cent6_forward_body", "", 1024 * 64),
    "cent6_trial_body": ("cent6_trial_body", "", 1024 * 101 * 64),  # round 5: stage merits of all line-search candidates
    "cent6_ls_body": ("cent6_ls_body", "", 1024 * 64),
    "id_quant_body": ("id_quant_body", "", 4096 * 64),
    "id_assemble_body": ("id_assemble_body", "", 4096 * 64),
    "qp_admm_body": ("qp_admm_body", "", 4096 * 64),
    "id6_assemble_body": ("id6_assemble_body", "", 4096 * 64),
    "qp6_admm_body": ("qp6_admm_body", "", 4096 * 64),
}


def kernel_key(name, grid_size):
    """Key of a full-batch main-symbol launch (TAG = 0), or None."""
    # TAG is the last template argument of kernel_entry: the symbol ends "...ELi<NT>ELi<MINW>ELi<TAG>EEEvS5_" (a robot shape that ends in
    # zeros carries the same characters in the middle of the symbol: look at the end only)
    if "ELi0EEEv" not in name.strip()[-24:]:
        return None
    for key, (sub, tag, grid) in KEYS.items():
        if sub in name and tag in name and int(grid_size) >= grid:
            return key
    return None
