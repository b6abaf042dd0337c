"""ctypes binding of libdurf_hip.so (C ABI in include/durf_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared
object is missing this module raises at import of any op (run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C durf_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DURF_LIB_PATH: kernel-tuning A/B builds (tools/build_variant.sh); the default is the in-tree build
LIB_PATH = os.environ.get('DURF_LIB_PATH') or os.path.join(_HERE, 'libdurf_hip.so')

_lib = None

vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float

_SIGS = {
    'durf_last_error': (C.c_char_p, []),
    'durf_version': (i32, []),
    'durf_mlp_param_count': (u64, [i32, i32]),
    'durf_mlp_layer_offset': (u64, [i32, i32, i32, i32]),
    'durf_wpack_fwd_bytes': (u64, [i32]),
    'durf_wpack_bwd_bytes': (u64, [i32]),
    'durf_pack_weights': (i32, [vp, i32, i32, vp, vp, vp]),
    'durf_pack_weights_all': (i32, [vp, vp, i32, vp, vp, i32, vp, u64, i32, vp, vp]),
    'durf_ray_setup': (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_compact_hits': (i32, [vp, i32, i32, vp, vp, vp, vp]),
    'durf_compact_classes': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    'durf_compact_all': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_ray_prologue': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    'durf_sample_t': (i32, [vp, i32, i32, vp, vp, vp, i32, vp]),
    'durf_view_enc': (i32, [vp, i32, vp, vp, vp]),
    'durf_encode_bkgd': (i32, [vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    'durf_encode_obj': (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32, vp, vp]),
    'durf_mlp_stash_bytes': (u64, [i32, u64]),
    'durf_mlp_mask_bytes': (u64, [u64]),
    'durf_mlp_fwd': (i32, [vp, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_composite_fwd': (i32, [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, f32, i32,
                                 vp, vp, vp, vp, vp, vp]),
    'durf_composite_resample': (i32, [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, f32, i32,
                                      vp, vp, vp, vp, vp, vp, f32, vp, vp,
                                      vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp, vp, vp]),
    'durf_resample': (i32, [vp, i32, i32, vp, vp, f32, vp, vp]),
    'durf_sorted_piecewise_constant_pdf': (i32, [vp, i32, i32, vp, vp, vp, vp]),
    'durf_mlp_f32_act_floats': (u64, [i32, i32]),
    'durf_mlp_f32_dz_floats': (u64, [i32, i32]),
    'durf_mlp_f32_dw_scratch_floats': (u64, [i32, i32, i32]),
    'durf_mlp_fwd_f32': (i32, [vp, i32, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp]),
    'durf_mlp_bwd_f32': (i32, [vp, i32, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp]),
    'durf_mlp_dw_f32': (i32, [vp, i32, i32, u64, i32, vp, vp, vp, i32, vp, vp, vp]),
    'durf_obj_enc_stride': (u64, [i32, i32]),
    'durf_obj_view_stride': (u64, [i32, i32]),
    'durf_obj_dzout_stride': (u64, [i32, i32]),
    'durf_pack_weights_batch': (i32, [vp, i32, i32, i32, vp, u64, vp, vp]),
    'durf_obj_fwd_batch': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32, vp, vp, vp, vp, vp, vp, vp]),
    'durf_obj_bwd_batch': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_obj_dw_batch': (i32, [vp, i32, i32, i32, vp, i32] + [C.POINTER(vp)] * 5 + [i32, vp, vp, vp, u64, vp]),
    'durf_obj_dw_partials': (i32, [vp, i32, i32, i32, vp, i32] + [C.POINTER(vp)] * 5 + [vp, vp]),
    'durf_dw_finalize_all': (i32, [vp, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp), vp, vp, vp, vp,
                                   i32, i32, i32, vp, i32, i32, vp, vp, vp, u64, vp]),
    'durf_gen_batch': (i32, [vp, i32, i32, C.POINTER(f32), vp, f32, f32, vp, vp, vp, i32] + [vp] * 10),
    'durf_ssim_scratch_floats': (u64, [i32, i32, i32, i32]),
    'durf_ssim': (i32, [vp, i32, i32, i32, vp, vp, f32, i32, vp, f32, f32, vp, vp, vp]),
    'durf_train_stats': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(f32), i32, vp,
                               C.POINTER(vp), i32]),
    'durf_loss_prep': (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp]),
    'durf_loss_bwd': (i32, [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                            f32, C.POINTER(f32), f32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_mlp_bwd': (i32, [vp, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'durf_expand_view': (i32, [vp, u64, i32, vp, vp, vp, vp, vp, vp]),
    'durf_dw_part_floats': (u64, [i32]),
    'durf_dw_bpart_floats': (u64, [i32]),
    'durf_mlp_dw': (i32, [vp, i32, u64, i32, vp, i32] + [C.POINTER(vp)] * 5 + [vp, vp]),
    'durf_mlp_dw_finalize': (i32, [vp, i32, i32, u64, i32, vp, i32, vp, vp, vp, vp]),
    'durf_mlp_dw_levels': (i32, [vp, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp)] + [C.POINTER(vp)] * 5 + [vp, vp]),
    'durf_mlp_dw_finalize_levels': (i32, [vp, i32, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp), vp, vp, vp, vp]),
    'durf_expand_raw': (i32, [vp, i32, i32, vp, vp, vp, vp]),
    'durf_encode_obj_bwd': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(f32), vp, vp]),
    'durf_encode_obj_bwd_batch': (i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(f32), vp, vp]),
    'durf_pose_finish': (i32, [vp, i32, vp, vp, i32, i32, vp]),
    'durf_optim_scratch_floats': (u64, [u64]),
    'durf_clip_adam': (i32, [vp, u64, vp, vp, vp, vp, f32, f32, f32, f32, i32, vp, vp]),
}


def symbols():
    """Every symbol include/durf_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libdurf_hip.so not built (%s): the HIP extension is required, there is no '
                'fallback path.  Run __graft_entry__.build().' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class DurfError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise DurfError('%s failed (%d): %s' % (what, rc, lib().durf_last_error().decode()))
