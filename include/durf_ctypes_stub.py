"""ctypes binding of libdurf_hip.so for a reference-side caller -- GENERATED from include/durf_hip.h by
tools/gen_integration_stub.py (do not edit; `--check` runs in the CPU test-suite).  All `vp` arguments are
DEVICE pointers of caller-owned buffers except `stream` (hipStream_t); C.POINTER(...) arguments are host
arrays.  Every call is asynchronous on `stream` and returns 0 or an error code (durf_last_error())."""
import ctypes as C

vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t


def bind(path='durf_amd/libdurf_hip.so'):
    L = C.CDLL(path)
    L.durf_last_error.restype = C.c_char_p
    L.durf_last_error.argtypes = []
    L.durf_version.restype = i32
    L.durf_version.argtypes = []
    L.durf_dispatch_seen.restype = i32
    L.durf_dispatch_seen.argtypes = []
    L.durf_dispatch_reset.restype = i32
    L.durf_dispatch_reset.argtypes = []
    L.durf_mlp_param_count.restype = u64
    L.durf_mlp_param_count.argtypes = [i32, i32]
    #   (width, in_dim)
    L.durf_mlp_layer_offset.restype = u64
    L.durf_mlp_layer_offset.argtypes = [i32, i32, i32, i32]
    #   (width, in_dim, layer, want_bias)
    L.durf_wpack_fwd_bytes.restype = u64
    L.durf_wpack_fwd_bytes.argtypes = [i32]
    #   (width)
    L.durf_wpack_bwd_bytes.restype = u64
    L.durf_wpack_bwd_bytes.argtypes = [i32]
    #   (width)
    L.durf_pack_weights.restype = i32
    L.durf_pack_weights.argtypes = [vp, i32, i32, vp, vp, vp]
    #   (stream, width, in_dim, mlp_params, wpack_fwd, wpack_bwd)
    L.durf_pack_weights_all.restype = i32
    L.durf_pack_weights_all.argtypes = [vp, vp, i32, vp, vp, i32, vp, u64, i32, vp, vp]
    #   (stream, bkgd_params, in_bkgd, bkgd_fwd, bkgd_bwd, K, obj_params, obj_param_stride, in_obj, obj_fwd, obj_bwd)
    L.durf_ray_setup.restype = i32
    L.durf_ray_setup.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, B, K, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo)
    L.durf_compact_hits.restype = i32
    L.durf_compact_hits.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    #   (stream, B, K, hit, idx, count, slot)
    L.durf_compact_classes.restype = i32
    L.durf_compact_classes.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp]
    #   (stream, B, K, N, hit, idx, count, slot, dyn)
    L.durf_compact_all.restype = i32
    L.durf_compact_all.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, B, K, N, hit, idx_obj, count_obj, slot_obj, idx_cls, count_cls, slot_cls, dyn)
    L.durf_ray_prologue.restype = i32
    L.durf_ray_prologue.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, u64, C.c_uint32, C.c_uint32, vp]
    #   (stream, B, K, N, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo, viewdirs, view_bf16, near, far, t_rand, lindisp, t_vals, pose_copy, zero_buf, zero_count, seed_lo, seed_hi, u_rand_out)
    L.durf_ray_prologue_pack.restype = i32
    L.durf_ray_prologue_pack.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, u64, C.c_uint32, C.c_uint32, vp, vp, i32, vp, vp, i32, vp, u64, i32, vp, vp, vp, u64]
    #   (stream, B, K, N, origins, dirs, pose, ext, origins_s, dirs_s, hit, zo, viewdirs, view_bf16, near, far, t_rand, lindisp, t_vals, pose_copy, zero_buf, zero_count, seed_lo, seed_hi, u_rand_out, bkgd_params, in_bkgd, bkgd_fwd, bkgd_bwd, K_pack, obj_params, obj_param_stride, in_obj, obj_fwd, obj_bwd, zero_buf2, zero_count2)
    L.durf_sample_t.restype = i32
    L.durf_sample_t.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp]
    #   (stream, B, N, near, far, t_rand, lindisp, t_vals)
    L.durf_density_noise.restype = i32
    L.durf_density_noise.argtypes = [vp, u64, vp, f32, vp, C.c_uint32, C.c_uint32, i32]
    #   (stream, rows, raw, scale, normal, seed_lo, seed_hi, level)
    L.durf_view_enc.restype = i32
    L.durf_view_enc.argtypes = [vp, i32, vp, vp, vp]
    #   (stream, B, viewdirs, out_bf16, out_f32)
    L.durf_encode_bkgd.restype = i32
    L.durf_encode_bkgd.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    #   (stream, B, N, t_vals, origins_s, dirs_s, radii, hit, K, contraction, out_tile, out_f32, idx, count)
    L.durf_encode_obj.restype = i32
    L.durf_encode_obj.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32, vp, vp]
    #   (stream, max_rays, N, idx, count, t_vals, origins_s, dirs_s, radii, barf_w, flags, out_tile, out_f32)
    L.durf_mlp_stash_bytes.restype = u64
    L.durf_mlp_stash_bytes.argtypes = [i32, u64]
    #   (width, rows)
    L.durf_mlp_mask_bytes.restype = u64
    L.durf_mlp_mask_bytes.argtypes = [u64]
    #   (rows)
    L.durf_mlp_fwd.restype = i32
    L.durf_mlp_fwd.argtypes = [vp, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, width, rows, N, enc_tile, view_bf16, ray_idx, count, wpack_fwd, raw, stash, relu_mask, tail_idx, tail_count)
    L.durf_mlp_fwd_enc.restype = i32
    L.durf_mlp_fwd_enc.argtypes = [vp, u64, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, rows, N, t_vals, origins_s, dirs_s, radii, hit, K, enc_flags, enc_tile, view_bf16, ray_idx, count, wpack_fwd, raw, stash, relu_mask, tail_idx, tail_count, view_tile)
    L.durf_mlp_fwd_enc_obj.restype = i32
    L.durf_mlp_fwd_enc_obj.argtypes = [vp, u64, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, C.POINTER(f32), i32, vp, vp, vp, vp, vp, vp]
    #   (stream, rows, N, t_vals, origins_s, dirs_s, radii, hit, K, enc_flags, enc_tile, view_bf16, ray_idx, count, wpack_fwd, raw, stash, relu_mask, tail_idx, tail_count, view_tile, B, obj_idx, obj_count, barf_w, obj_flags, obj_wpack_fwd, obj_enc, obj_raw, obj_stash, obj_relu_mask, obj_view_tile)
    L.durf_composite_fwd.restype = i32
    L.durf_composite_fwd.argtypes = [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp]
    #   (stream, B, N, K, raw_bkgd, raw_obj, slot, t_vals, dirs_s, density_bias, bkgd_mode, rgb, depth, acc, weights, t_mids, t_dists)
    L.durf_composite_resample.restype = i32
    L.durf_composite_resample.argtypes = [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp, vp, vp]
    #   (stream, B, N, K, raw_bkgd, raw_obj, slot, t_vals, dirs_s, density_bias, bkgd_mode, rgb, depth, acc, weights, t_mids, t_dists, resample_padding, u_rand, t_vals_out, lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, level, disable_multiscale, prep_this, norm_this, prep_next, norm_next)
    L.durf_resample.restype = i32
    L.durf_resample.argtypes = [vp, i32, i32, vp, vp, f32, vp, vp]
    #   (stream, B, N, t_vals, weights, resample_padding, u_rand, t_vals_out)
    L.durf_sorted_piecewise_constant_pdf.restype = i32
    L.durf_sorted_piecewise_constant_pdf.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    #   (stream, B, N, bins, weights, u_rand, samples)
    L.durf_loss_prep.restype = i32
    L.durf_loss_prep.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp]
    #   (stream, B, N, t_vals, lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, level, disable_multiscale, prep, norm)
    L.durf_loss_bwd.restype = i32
    L.durf_loss_bwd.argtypes = [vp, i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, C.POINTER(f32), f32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, B, N, K, raw_bkgd, raw_obj, slot, t_vals, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo, norm, eps, mults, box_loss_mult, level, disable_multiscale, bg, density_bias, draw, terms, term_sums, rgb_out, depth_out, acc_out, weights_out, t_mids_out, t_dists_out, draw_ray_sum)
    L.durf_loss_bwd_levels.restype = i32
    L.durf_loss_bwd_levels.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, f32, f32]
    #   (stream, B, N, K, L, levels, slot, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, disable_multiscale, bg, density_bias)
    L.durf_train_stats.restype = i32
    L.durf_train_stats.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(f32), i32, vp, C.POINTER(vp), i32]
    #   (stream, L, K, N, norms, sums, weight_l2, pose6, prev6, target6, t_vals, mults, mode, out, terms, B)
    L.durf_mlp_bwd.restype = i32
    L.durf_mlp_bwd.argtypes = [vp, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, width, rows, N, draw, ray_idx, count, wpack_bwd, relu_mask, dz, dz_out, d_enc, tail_idx, tail_count, draw_ray_sum)
    L.durf_expand_view.restype = i32
    L.durf_expand_view.argtypes = [vp, u64, i32, vp, vp, vp, vp, vp, vp]
    #   (stream, rows, N, view_bf16, ray_idx, count, out_tile, tail_idx, tail_count)
    L.durf_dw_part_floats.restype = u64
    L.durf_dw_part_floats.argtypes = [i32]
    #   (width)
    L.durf_dw_bpart_floats.restype = u64
    L.durf_dw_bpart_floats.argtypes = [i32]
    #   (width)
    L.durf_mlp_dw.restype = i32
    L.durf_mlp_dw.argtypes = [vp, i32, u64, i32, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]
    #   (stream, width, rows, N, count, nlevels, enc_tile, view_tile, stash, dz, dz_out, part, bpart)
    L.durf_mlp_dw_finalize.restype = i32
    L.durf_mlp_dw_finalize.argtypes = [vp, i32, i32, u64, i32, vp, i32, vp, vp, vp, vp]
    #   (stream, width, in_dim, rows, N, count, nlevels, part, bpart, grad_mlp, mlp_params)
    L.durf_mlp_dw_levels.restype = i32
    L.durf_mlp_dw_levels.argtypes = [vp, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]
    #   (stream, width, nlevels, rows, rows_per_ray, count, enc_tile, view_tile, stash, dz, dz_out, part, bpart)
    L.durf_mlp_dw_finalize_levels.restype = i32
    L.durf_mlp_dw_finalize_levels.argtypes = [vp, i32, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp), vp, vp, vp, vp]
    #   (stream, width, in_dim, nlevels, rows, rows_per_ray, count, part, bpart, grad_mlp, mlp_params)
    L.durf_expand_raw.restype = i32
    L.durf_expand_raw.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    #   (stream, B, N, raw_c, count, slot, raw_full, raw_tail)
    L.durf_forward_workspace_bytes.restype = u64
    L.durf_forward_workspace_bytes.argtypes = [i32, i32, i32]
    #   (B, N, K)
    L.durf_forward.restype = i32
    L.durf_forward.argtypes = [vp, vp, vp, u64]
    #   (stream, args, workspace, workspace_bytes)
    L.durf_render_image_workspace_bytes.restype = u64
    L.durf_render_image_workspace_bytes.argtypes = [i32, i32, i32, i32]
    #   (chunk, N, K, num_levels)
    L.durf_render_image.restype = i32
    L.durf_render_image.argtypes = [vp, vp, u64, i32, vp, vp, vp, vp, u64]
    #   (stream, args, n_rays, chunk, rgb, distance, acc, workspace, workspace_bytes)
    L.durf_train_workspace_bytes.restype = u64
    L.durf_train_workspace_bytes.argtypes = [i32, i32, i32, i32, u64]
    #   (B, N, K, num_levels, n_params)
    L.durf_train_workspace_bytes_flags.restype = u64
    L.durf_train_workspace_bytes_flags.argtypes = [i32, i32, i32, i32, u64, i32]
    #   (B, N, K, num_levels, n_params, flags)
    L.durf_prefetch_join.restype = i32
    L.durf_prefetch_join.argtypes = [vp]
    #   (stream)
    L.durf_loss_backward.restype = i32
    L.durf_loss_backward.argtypes = [vp, vp, vp, u64]
    #   (stream, args, workspace, workspace_bytes)
    L.durf_train_step.restype = i32
    L.durf_train_step.argtypes = [vp, vp, vp, u64]
    #   (stream, args, workspace, workspace_bytes)
    L.durf_comm_available.restype = i32
    L.durf_comm_available.argtypes = []
    L.durf_comm_unique_id.restype = i32
    L.durf_comm_unique_id.argtypes = [vp]
    #   (id_out)
    L.durf_comm_init.restype = i32
    L.durf_comm_init.argtypes = [i32, i32, vp, C.POINTER(vp)]
    #   (world, rank, id, comm_out)
    L.durf_comm_destroy.restype = i32
    L.durf_comm_destroy.argtypes = [vp]
    #   (comm)
    L.durf_allreduce_sum.restype = i32
    L.durf_allreduce_sum.argtypes = [vp, vp, vp, u64]
    #   (stream, comm, buf, n)
    L.durf_mlp_f32_act_floats.restype = u64
    L.durf_mlp_f32_act_floats.argtypes = [i32, i32]
    #   (width, in_dim)
    L.durf_mlp_f32_dz_floats.restype = u64
    L.durf_mlp_f32_dz_floats.argtypes = [i32, i32]
    #   (width, in_dim)
    L.durf_mlp_f32_dw_scratch_floats.restype = u64
    L.durf_mlp_f32_dw_scratch_floats.argtypes = [i32, i32, i32]
    #   (width, in_dim, nsplit)
    L.durf_mlp_f32_wstream_floats.restype = u64
    L.durf_mlp_f32_wstream_floats.argtypes = [i32]
    #   (width)
    L.durf_mlp_f32_pack.restype = i32
    L.durf_mlp_f32_pack.argtypes = [vp, i32, i32, i32, vp, u64, vp]
    #   (stream, width, in_dim, K, mlp_params, param_stride, wstream)
    L.durf_mlp_fwd_f32.restype = i32
    L.durf_mlp_fwd_f32.argtypes = [vp, i32, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, width, in_dim, rows, N, enc, view27, ray_idx, count, mlp_params, wstream, raw, act)
    L.durf_mlp_bwd_f32.restype = i32
    L.durf_mlp_bwd_f32.argtypes = [vp, i32, i32, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, width, in_dim, rows, N, draw, ray_idx, count, mlp_params, wstream, act, dz, d_enc)
    L.durf_mlp_dw_f32.restype = i32
    L.durf_mlp_dw_f32.argtypes = [vp, i32, i32, u64, i32, vp, vp, vp, i32, vp, vp]
    #   (stream, width, in_dim, rows, N, count, act, dz, nsplit, scratch, grad_mlp)
    L.durf_bkgd_const_trunk_f32.restype = i32
    L.durf_bkgd_const_trunk_f32.argtypes = [vp, vp, vp]
    #   (stream, bkgd_params, trunk)
    L.durf_bkgd_hit_rays_f32.restype = i32
    L.durf_bkgd_hit_rays_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    #   (stream, B, view27, bkgd_params, idx, count, trunk, raw_tail)
    L.durf_objf32_act_stride.restype = u64
    L.durf_objf32_act_stride.argtypes = [i32, i32]
    #   (B, N)
    L.durf_objf32_dz_stride.restype = u64
    L.durf_objf32_dz_stride.argtypes = [i32, i32]
    #   (B, N)
    L.durf_encode_obj_f32_batch.restype = i32
    L.durf_encode_obj_f32_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32, vp]
    #   (stream, K, B, N, idx, count, t_vals, origins_s, dirs_s, radii, barf_w, flags, enc)
    L.durf_objf32_fwd_batch.restype = i32
    L.durf_objf32_fwd_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32]
    #   (stream, K, B, N, idx, count, enc, view27, obj_params, param_stride, wstream, raw, act, t_vals, origins_s, dirs_s, radii, barf_w, flags)
    L.durf_objf32_bwd_batch.restype = i32
    L.durf_objf32_bwd_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, u64, vp, vp, vp, vp]
    #   (stream, K, B, N, idx, count, draw, obj_params, param_stride, wstream, act, dz, d_enc)
    L.durf_mlp_f32_pack_x3.restype = i32
    L.durf_mlp_f32_pack_x3.argtypes = [vp, i32, vp, u64, vp]
    #   (stream, K, obj_params, param_stride, wstream)
    L.durf_objf32_bwd_batch_x3.restype = i32
    L.durf_objf32_bwd_batch_x3.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, u64, vp, vp, vp, vp]
    #   (stream, K, B, N, idx, count, draw, obj_params, param_stride, wstream, act, dz, d_enc)
    L.durf_objf32_dw_batch.restype = i32
    L.durf_objf32_dw_batch.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(vp), C.POINTER(vp), i32, vp, vp, u64]
    #   (stream, K, B, N, count, nlevels, act, dz, nsplit, scratch, grad_obj, grad_stride)
    L.durf_obj_enc_stride.restype = u64
    L.durf_obj_enc_stride.argtypes = [i32, i32]
    #   (B, N)
    L.durf_obj_view_stride.restype = u64
    L.durf_obj_view_stride.argtypes = [i32, i32]
    #   (B, N)
    L.durf_obj_dzout_stride.restype = u64
    L.durf_obj_dzout_stride.argtypes = [i32, i32]
    #   (B, N)
    L.durf_pack_weights_batch.restype = i32
    L.durf_pack_weights_batch.argtypes = [vp, i32, i32, i32, vp, u64, vp, vp]
    #   (stream, width, in_dim, K, mlp_params, param_stride, wpack_fwd, wpack_bwd)
    L.durf_obj_fwd_batch.restype = i32
    L.durf_obj_fwd_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(f32), i32, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, K, B, N, idx, count, t_vals, origins_s, dirs_s, radii, barf_w, flags, view_bf16, wpack_fwd, enc, raw, stash, relu_mask, view_tile)
    L.durf_obj_bwd_batch.restype = i32
    L.durf_obj_bwd_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, K, B, N, idx, count, draw, wpack_bwd, relu_mask, dz, dz_out, d_enc)
    L.durf_obj_bwd_batch_levels.restype = i32
    L.durf_obj_bwd_batch_levels.argtypes = [vp, i32, i32, i32, i32, vp, vp, C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    #   (stream, K, B, N, nlevels, idx, count, draw, wpack_bwd, relu_mask, dz, dz_out)
    L.durf_mlp_bwd_obj.restype = i32
    L.durf_mlp_bwd_obj.argtypes = [vp, u64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    #   (stream, rows, N, draw, ray_idx, count, wpack_bwd, relu_mask, dz, dz_out, tail_idx, tail_count, draw_ray_sum, K, B, nlevels, obj_idx, obj_count, obj_draw, obj_wpack_bwd, obj_relu_mask, obj_dz, obj_dz_out)
    L.durf_obj_dw_batch.restype = i32
    L.durf_obj_dw_batch.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, vp, vp, vp, u64, vp]
    #   (stream, K, B, N, count, nlevels, enc, view_tile, stash, dz, dz_out, in_dim, part, bpart, grad_mlp, grad_stride, mlp_params)
    L.durf_obj_dw_partials.restype = i32
    L.durf_obj_dw_partials.argtypes = [vp, i32, i32, i32, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]
    #   (stream, K, B, N, count, nlevels, enc, view_tile, stash, dz, dz_out, part, bpart)
    L.durf_dw_finalize_all.restype = i32
    L.durf_dw_finalize_all.argtypes = [vp, i32, i32, C.POINTER(u64), C.POINTER(i32), C.POINTER(vp), vp, vp, vp, vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, u64, vp]
    #   (stream, in_bkgd, nseg, rows, rows_per_ray, seg_count, part_bkgd, bpart_bkgd, grad_bkgd, bkgd_params, K, B, N, obj_count, nlevels, in_obj, part_obj, bpart_obj, grad_obj, obj_grad_stride, obj_params)
    L.durf_encode_obj_bwd.restype = i32
    L.durf_encode_obj_bwd.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(f32), vp, vp, i32, i32]
    #   (stream, B, N, k_obj, idx, count, d_enc, t_vals, origins_s, dirs_s, radii, origins, dirs, pose, barf_w, scratch, sums, precise, enc_flags)
    L.durf_encode_obj_bwd_batch.restype = i32
    L.durf_encode_obj_bwd_batch.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(f32), vp, vp, i32, i32]
    #   (stream, K, B, N, idx, count, d_enc, t_vals, origins_s, dirs_s, radii, origins, dirs, pose, barf_w, scratch, sums, precise, enc_flags)
    L.durf_encode_obj_bwd_levels.restype = i32
    L.durf_encode_obj_bwd_levels.argtypes = [vp, i32, i32, i32, i32, vp, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp, vp, vp, C.POINTER(f32), vp, vp, i32, i32]
    #   (stream, K, B, N, nlevels, idx, count, d_enc, t_vals, origins_s, dirs_s, radii, origins, dirs, pose, barf_w, scratch, sums, precise, enc_flags)
    L.durf_pose_finish.restype = i32
    L.durf_pose_finish.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    #   (stream, K, pose, sums, want_pos, want_rot, grad6)
    L.durf_optim_scratch_floats.restype = u64
    L.durf_optim_scratch_floats.argtypes = [u64]
    #   (n)
    L.durf_stats_scrub.restype = i32
    L.durf_stats_scrub.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(f32), i32, vp, C.POINTER(vp), i32, u64, vp, f32, f32, vp, vp, u64, i32, u64, u64]
    #   (stream, L, K, N, norms, sums, weight_l2, pose6, prev6, target6, t_vals, mults, mode, out, terms, B, n, grad, inv_world, max_val, scratch, cls_count, box_floats, K_boxes, mlp0_floats, obj_floats)
    L.durf_adam_apply.restype = i32
    L.durf_adam_apply.argtypes = [vp, u64, vp, vp, vp, vp, f32, f32, i32, vp, vp]
    #   (stream, n, params, m, v, grad, max_norm, lr, step, scratch, stats)
    L.durf_poison_multi_hit.restype = i32
    L.durf_poison_multi_hit.argtypes = [vp, u64, vp, vp, u64, i32, u64, u64]
    #   (stream, n, grad, cls_count, box_floats, K, mlp0_floats, obj_floats)
    L.durf_clip_adam.restype = i32
    L.durf_clip_adam.argtypes = [vp, u64, vp, vp, vp, vp, f32, f32, f32, f32, i32, vp, vp]
    #   (stream, n, params, m, v, grad, inv_world, max_val, max_norm, lr, step, scratch, stats)
    L.durf_weight_decay.restype = i32
    L.durf_weight_decay.argtypes = [vp, u64, vp, vp, u64, u64, f32, vp, vp]
    #   (stream, n, params, grad, lo, hi, mult, scratch, weight_l2)
    L.durf_gen_batch.restype = i32
    L.durf_gen_batch.argtypes = [vp, i32, i32, C.POINTER(f32), vp, f32, f32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    #   (stream, B, n_cams, cams_host, ray_idx, near, far, images, depth, sky, img_channels, origins, directions, viewdirs, radii, lossmult, near_out, far_out, pixels, depth_out, sky_out)
    L.durf_ssim_scratch_floats.restype = u64
    L.durf_ssim_scratch_floats.argtypes = [i32, i32, i32, i32]
    #   (H, W, C, filter_size)
    L.durf_ssim.restype = i32
    L.durf_ssim.argtypes = [vp, i32, i32, i32, vp, vp, f32, i32, vp, f32, f32, vp, vp, vp]
    #   (stream, H, W, C, img0, img1, max_val, filter_size, filt_dev, k1, k2, ssim_map, scratch, ssim_mean)
    return L
