"""ctypes binding of libpyfem_hip.so (C ABI: include/pyfem_hip.h).

There is no CPU fallback: if the library is missing or cannot be loaded the import of any
compute entry point raises -- the product path never routes through a numpy stand-in.
Status codes are turned into the exception types the reference raises at the same places
(assert -> AssertionError for mesh checks, pyfem.py:680-681; RuntimeError for a failed
iterative solve, pyfem.py:2421-2422; ValueError for unsupported element types, pyfem.py:2715).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PF_LIB_PATH: tools/ point this at libpyfem_hip_exp.so (the -DPF_EXPERIMENTS build with the ablation / tuning knobs)
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "libpyfem_hip.so")

PF_OK, PF_E_INVALID, PF_E_HIP, PF_E_UNSUPPORTED, PF_E_NOCONV = 0, -1, -2, -3, -4
PF_SCATTER_GATHER, PF_SCATTER_ATOMIC = 0, 1
PF_PRECOND_NONE, PF_PRECOND_JACOBI, PF_PRECOND_AMG, PF_PRECOND_AMG_REUSE, PF_PRECOND_AMG_STALE = 0, 1, 2, 3, 4
PF_K_ASSEMBLE_ELASTICITY, PF_K_SPMV, PF_K_ASSEMBLE_HELMHOLTZ = 0, 1, 2
PF_HBM_FILL, PF_HBM_READ, PF_HBM_MIX = 0, 1, 2
PF_TAB_ELASTICITY, PF_TAB_POISSON = 0, 1
PF_NL_TANGENT, PF_NL_TANGENT_SPD, PF_NL_ELEMENT_TANGENT, PF_NL_RESIDUAL = 0, 1, 2, 3

_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); must list every symbol include/pyfem_hip.h declares
SIGNATURES = {
    "pf_version": (C.c_int, []),
    "pf_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pf_ctx_destroy": (C.c_int, [_vp]),
    "pf_last_error": (C.c_char_p, [_vp]),
    "pf_last_warning": (C.c_char_p, [_vp]),
    "pf_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "pf_mesh_set": (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, _vp, C.c_int]),
    "pf_mesh_set3d": (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, _vp, C.c_int]),
    "pf_pattern_build": (C.c_int, [_vp, C.c_int]),
    "pf_pattern_dims": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "pf_pattern_get_graph": (C.c_int, [_vp, _vp, _vp]),
    "pf_pattern_get_csr": (C.c_int, [_vp, _vp, _vp]),
    "pf_pattern_get_compact": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64), _vp]),
    "pf_pattern_get_tiles": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64), _vp, _vp, _vp]),
    "pf_pattern_tile_info": (C.c_int, [_vp, C.POINTER(_i32)]),
    "pf_assemble_elasticity2d": (C.c_int, [_vp, _f64, _f64, _f64, _vp, C.c_int, _vp]),
    "pf_assemble_helmholtz2d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_sens_elasticity2d": (C.c_int, [_vp, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "pf_assemble_poisson2d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_assemble_elasticity3d": (C.c_int, [_vp, _f64, _f64, _f64, _vp, _vp]),
    "pf_assemble_poisson3d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_assemble_helmholtz3d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_element_matrices3d": (C.c_int, [_vp, C.c_int, _f64, _f64, _f64, _f64, _vp, _vp, _vp]),
    "pf_sens3d": (C.c_int, [_vp, C.c_int, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "pf_assemble_tab": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _f64, _f64, _vp, _vp, _vp]),
    "pf_element_matrices_tab": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _f64, _f64, _vp, _vp, _vp]),
    "pf_sens_tab": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "pf_quad_points3d": (C.c_int, [_vp, _vp]),
    "pf_poisson_rhs3d": (C.c_int, [_vp, _vp, _vp]),
    "pf_element_matrices_poisson2d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_quad_points": (C.c_int, [_vp, _vp]),
    "pf_poisson_rhs": (C.c_int, [_vp, _vp, _vp]),
    "pf_sens_poisson2d": (C.c_int, [_vp, _f64, _vp, _vp, _vp, _vp]),
    "pf_element_matrices_elasticity2d": (C.c_int, [_vp, _f64, _f64, _f64, _vp, _vp]),
    "pf_element_matrices_helmholtz2d": (C.c_int, [_vp, _f64, _vp, _vp]),
    "pf_vals_to_csr": (C.c_int, [_vp, _vp, _vp]),
    "pf_apply_dirichlet": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "pf_spmv": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pf_extract_diag": (C.c_int, [_vp, _vp, _vp]),
    "pf_cg": (C.c_int, [_vp, _vp, _vp, _vp, _f64, _f64, C.c_int, C.c_int, C.c_int, C.POINTER(_i32),
                        C.POINTER(_f64)]),
    "pf_cg2": (C.c_int, [_vp, _vp, _vp, _vp, _f64, _f64, C.c_int, C.c_int, C.POINTER(_i32), C.POINTER(_f64)]),
    "pf_amg_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_f64), C.c_int]),
    "pf_amg_prepare": (C.c_int, [_vp, _vp, C.c_int]),
    "pf_amg_set_power_steps": (C.c_int, [_vp, C.c_int]),
    "pf_bicgstab": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f64, _f64, C.c_int, C.c_int, C.POINTER(_i32), C.POINTER(_f64)]),
    "pf_nlpoisson_tab": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "pf_amg_vcycle": (C.c_int, [_vp, _vp, _vp]),
    "pf_add_to_diag": (C.c_int, [_vp, _vp, _vp]),
    "pf_dot": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "pf_copy_indexed": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "pf_spmv_dot": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pf_cgcg_update_xrz": (C.c_int, [_vp] * 9),
    "pf_cgcg_reduce": (C.c_int, [_vp, _vp, _vp, _i64]),
    "pf_cgcg_update_ps": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "pf_amg_level0": (C.c_int, [_vp, _vp, _vp, C.POINTER(_i64)]),
    "pf_amg_restrict0": (C.c_int, [_vp, _vp, _vp]),
    "pf_amg_coarse_solve": (C.c_int, [_vp, _vp, _vp]),
    "pf_amg_prolong0": (C.c_int, [_vp, _vp, _vp]),
    "pf_amg_p0_pattern": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp]),
    "pf_amg_prolong0_rows": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "pf_amg_restrict0_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_amg_p0_gather": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pf_amg_restrict0_stream": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_vec_smooth": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "pf_halo_create": (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.POINTER(_vp)]),
    "pf_halo_handle": (C.c_int, [_vp, _vp]),
    "pf_halo_connect": (C.c_int, [_vp, C.c_int, _vp]),
    "pf_halo_plan_create": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "pf_halo_exchange": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "pf_halo_status": (C.c_int, [_vp, C.POINTER(_i32)]),
    "pf_halo_set_timeout": (C.c_int, [_vp, _f64]),
    "pf_halo_destroy": (C.c_int, [_vp]),
    "pf_halo_plan_fuse": (C.c_int, [_vp, C.c_int, _vp]),
    "pf_halo_fuse_get": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pf_halo_set_colocated": (C.c_int, [_vp, C.c_int]),
    "pf_dist_spmv": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "pf_host_fuse_tables_c": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _vp]),
    "pf_dist_cg": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _f64, _f64, C.c_int, C.c_int, C.POINTER(_i32),
                             C.POINTER(_f64)]),
    "pf_dist_vcycle": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pf_vals_to_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pf_dist_pcg_amg": (C.c_int, [_vp, _vp, _vp, _vp, _f64, _f64, C.c_int, C.c_int, C.POINTER(_i32), C.POINTER(_f64)]),
    "pf_time_kernel": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _f64, _f64, _f64, C.POINTER(_f64)]),
    "pf_hbm_probe": (C.c_int, [_vp, C.c_int, C.c_int, _i64, C.c_int, C.POINTER(_f64)]),
}



class DistAmgArgs(C.Structure):
    """pf_dist_amg of include/pyfem_hip.h."""
    _fields_ = [("local", _vp), ("global_", _vp), ("halo", _vp), ("halo_coarse", _vp), ("plan_dof", _i32), ("plan_coarse", _i32),
                ("K", _vp), ("own_w", _vp), ("d1", _vp), ("d2", _vp), ("nl", _i64), ("grow", _vp), ("ltptr", _vp),
                ("ltnode", _vp), ("ltslot", _vp), ("n1", _i64), ("K32", _vp), ("ltvals", _vp)]


_lib = None


class PyfemHipError(RuntimeError):
    pass


def load():
    """dlopen the library and declare every signature.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    # torch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  If this library (linked
    # against the system ROCm) is dlopen'ed first, the process ends up with two HSA runtimes and the
    # second one sees no device.  Loading torch first makes both share torch's copy.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m pyfem_gpu_testflight_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for the accelerated path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc, what=""):
    """Map a status code to the reference's exception behaviour."""
    if rc == PF_OK:
        return
    msg = load().pf_last_error(ctx)
    msg = msg.decode() if msg else f"{what}: error {rc}"
    if rc == PF_E_NOCONV:
        raise RuntimeError(msg)  # pyfem.py:2421-2422
    if rc == PF_E_UNSUPPORTED:
        raise ValueError(msg)  # pyfem.py:2715
    if rc == PF_E_INVALID and "conn must span" in msg:
        raise AssertionError(msg)  # pyfem.py:680-681
    raise PyfemHipError(msg)
