"""ctypes declarations for libbridgehip.so (the C ABI of include/bridgehip.h).

The shared library is the product; this module only loads it.  There is NO CPU fallback: if the
library is missing the import fails, and creating a context without a GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BRIDGEHIP_SO lets kernel experiments (A/B builds of the same library) be benchmarked side by side
SO_PATH = os.environ.get("BRIDGEHIP_SO") or os.path.join(_HERE, "libbridgehip.so")

dp = C.POINTER(C.c_double)
vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/bridgehip.h
SIGNATURES = {
    "bhip_version": (C.c_int, []),
    "bhip_device_count": (C.c_int, []),
    "bhip_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "bhip_ctx_destroy": (None, [vp]),
    "bhip_ctx_sync": (C.c_int, [vp]),
    "bhip_ctx_set_option": (C.c_int, [vp, C.c_int, C.c_int]),
    "bhip_ctx_get_option": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "bhip_last_error": (C.c_char_p, [vp]),
    "bhip_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "bhip_free": (C.c_int, [vp, vp]),
    "bhip_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "bhip_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "bhip_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "bhip_upload_aos": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, dp]),
    "bhip_download_aos": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, dp]),
    "bhip_model_define": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "bhip_model_define_components": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "bhip_model_define_sigma": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "bhip_proposal_create": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.POINTER(vp)]),
    "bhip_proposal_destroy": (None, [vp]),
    "bhip_proposal_set_aux": (C.c_int, [vp, C.c_int, dp, C.c_int]),
    "bhip_proposal_set_aux_callback": (C.c_int, [vp, vp, vp, C.c_int, dp]),
    "bhip_proposal_set_aux_linearappr": (C.c_int, [vp, dp, dp, dp, dp]),
    "bhip_linearappr": (C.c_int, [vp, dp, dp, dp, dp]),
    "bhip_linearnoiseappr_path": (C.c_int, [vp, dp, C.c_int, dp]),
    "bhip_proposal_set_aux_linearnoiseappr": (C.c_int, [vp, dp]),
    "bhip_proposal_guide_hv": (C.c_int, [vp, dp, dp]),
    "bhip_proposal_guide_lmmu": (C.c_int, [vp, C.c_int, dp, dp, dp]),
    "bhip_proposal_guide_nuh": (C.c_int, [vp, C.c_int, dp, dp, C.c_double, dp, C.c_int]),
    "bhip_proposal_guide_arrays": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, dp, dp]),
    "bhip_proposal_guide_get": (C.c_int, [vp, dp, dp, dp, dp]),
    "bhip_proposal_lptilde": (C.c_int, [vp, dp, dp]),
    "bhip_proposal_info": (C.c_int, [vp] + [C.POINTER(C.c_int)] * 5),
    "bhip_wiener_sample": (C.c_int, [vp, dp, C.c_int, C.c_int, vp, C.c_long, C.c_long, C.c_uint64, C.c_uint32, C.c_uint32]),
    "bhip_solve": (C.c_int, [vp, vp, dp, vp, vp, C.c_long, vp, C.c_long, vp, C.c_int, C.c_long]),
    "bhip_sample_solve": (C.c_int, [vp, vp, dp, vp, vp, C.c_long, vp, C.c_long, vp, C.c_int, C.c_long,
                                    C.c_uint64, C.c_uint32, C.c_uint32]),
    "bhip_wiener_sample_parts": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.c_long, C.c_long, C.c_long, C.c_uint64, C.c_uint32, C.c_uint32]),
    "bhip_solve_parts": (C.c_int, [vp, vp, dp, C.c_int, C.POINTER(vp), C.c_long, C.c_long, C.c_int, C.POINTER(vp), C.c_long, C.c_long, vp, C.c_int, C.c_long]),
    "bhip_girsanov_parts": (C.c_int, [vp, vp, dp, C.c_int, C.c_int, C.POINTER(vp), C.c_long, C.c_long, vp, C.c_long]),
    "bhip_llikelihood_parts": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp), C.c_long, C.c_long, vp, C.c_int, C.c_long]),
    "bhip_sample_solve_parts": (C.c_int, [vp, vp, dp, C.c_int, C.POINTER(vp), C.c_long, C.c_long, vp, C.c_int, C.c_long,
                                          C.c_uint64, C.c_uint32, C.c_uint32]),
    "bhip_alloc_apart": (C.c_int, [vp, C.c_int, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_int)]),
    "bhip_free_apart": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "bhip_llikelihood": (C.c_int, [vp, vp, vp, C.c_long, vp, C.c_int, C.c_long]),
    "bhip_innovations": (C.c_int, [vp, vp, vp, C.c_long, vp, C.c_long, C.c_long]),
    "bhip_girsanov": (C.c_int, [vp, vp, dp, C.c_int, vp, C.c_long, vp, C.c_long]),
    "bhip_gpupdate": (C.c_int, [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp]),
    "bhip_chains_create": (C.c_int, [vp, vp, C.c_long, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "bhip_chains_destroy": (None, [vp]),
    "bhip_chains_init": (C.c_int, [vp, dp, C.c_int]),
    "bhip_chains_step": (C.c_int, [vp, C.c_double, C.c_int, C.c_int]),
    "bhip_chains_step_group": (C.c_int, [C.c_int, C.POINTER(vp), C.c_double, C.c_int, C.c_int]),
    "bhip_chains_stats_group": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "bhip_chains_iterations": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
    "bhip_chains_placement_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bhip_chains_placement_pieces": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bhip_ctx_piece_of": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_int)]),
    "bhip_chains_stats": (C.c_int, [vp, vp]),
    "bhip_chains_get": (C.c_int, [vp, dp, C.POINTER(C.c_int64)]),
    "bhip_chains_get_paths": (C.c_int, [vp, C.c_long, C.c_long, dp, dp]),
    "bhip_chains_current_X": (C.c_int, [vp, vp, C.c_long]),
    "bhip_chains_proposal_X": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_long)]),
    "bhip_chains_pathstats": (C.c_int, [vp, dp, dp]),
    "bhip_chains_state_bytes": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "bhip_chains_save": (C.c_int, [vp, vp]),
    "bhip_chains_load": (C.c_int, [vp, vp]),
    "bhip_welford_merge": (C.c_int, [C.c_long, C.c_int, dp, dp, dp, C.c_double, dp, dp]),
    "bhip_segchains_create": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_long, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "bhip_segchains_destroy": (None, [vp]),
    "bhip_segchains_init": (C.c_int, [vp, dp, dp, C.c_int]),
    "bhip_segchains_step": (C.c_int, [vp, dp, dp, C.c_int]),
    "bhip_segchains_placement_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bhip_segchains_statistics_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bhip_segchains_get": (C.c_int, [vp, dp, C.POINTER(C.c_int64), dp]),
    "bhip_segchains_get_paths": (C.c_int, [vp, C.c_int, C.c_long, C.c_long, dp, dp]),
    "bhip_segchains_current_X": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_long)]),
    "bhip_segchains_mcstats": (C.c_int, [vp, C.c_int, C.c_long, dp, dp, C.POINTER(C.c_int64)]),
    "bhip_segchains_set_proposals": (C.c_int, [vp, C.POINTER(vp)]),
    "bhip_segchains_pooled_stats": (C.c_int, [vp, C.c_int, dp, dp, dp]),
    "bhip_segchains_set_pi0": (C.c_int, [vp, dp, dp, C.c_int]),
    "bhip_segchains_adapt_device": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int]),
    "bhip_segchains_chain_guide": (C.c_int, [vp, C.c_int, C.c_long, dp, dp, dp]),
    "bhip_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
    "bhip_comm_init_rank": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "bhip_comm_init_all": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "bhip_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bhip_comm_query": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bhip_comm_allgather": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "bhip_comm_allgather_stats": (C.c_int, [vp, vp, vp]),
    "bhip_comm_init": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "bhip_allgather_stats": (C.c_int, [vp, vp, vp]),
    "bhip_comm_allgather_group": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_size_t]),
    "bhip_comm_destroy": (None, [vp]),
    "bhip_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bhip_normals_host": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, dp]),
    "bhip_normals_host_spec": (None, [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, dp]),
}

AUX_FN = C.CFUNCTYPE(None, C.c_double, dp, dp, dp, vp)

_lib = None


def load():
    """dlopen libbridgehip.so and attach prototypes.  torch is imported first so that this process
    uses ONE HIP runtime (PyTorch-ROCm bundles libamdhip64 under the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (device memory / streams / torch.distributed plumbing)
    except Exception:  # pragma: no cover - torch is part of the image
        pass
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build it with `python __graft_entry__.py` "
                          f"(or make -C bridge.jl_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(SO_PATH)
    old_build = os.environ.get("BRIDGEHIP_SO_OLD_BUILD") == "1"   # scripts/gpu_ab_*.sh: an OLDER library build for same-box A/B timing
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        except AttributeError:
            if old_build:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
