"""ctypes binding of librdm_hip.so (the C ABI in include/rdm_hip.h).

The product path has NO fallback: if the library is missing or a call fails, this raises.
PyTorch only provides device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librdm_hip.so")

c_f32p = C.c_void_p  # all device pointers travel as void*
i32, i64, f32, f64, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t


class ConvDesc(C.Structure):
    """rdm_conv_desc"""
    _fields_ = [(n, i32) for n in ("batch", "in_h", "in_w", "in_c", "in_ld", "out_c", "out_ld", "kh", "kw", "stride_h", "stride_w", "pad_h", "pad_w")]


_SIGNATURES = {
    # name: (restype, [argtypes])
    "rdm_last_error_string": (C.c_char_p, []),
    "rdm_version": (C.c_int, []),
    "rdm_profile_enable": (None, [i32]),
    "rdm_launch_count": (i64, []),
    "rdm_census_enable": (None, [i32]),
    "rdm_census_reset": (None, []),
    "rdm_census_count": (i32, []),
    "rdm_census_entry": (C.c_int, [i32, C.POINTER(C.c_char_p), C.POINTER(i64)]),
    "rdm_profile_read": (C.c_int, [C.POINTER(f64), C.POINTER(f64), C.POINTER(f64), C.POINTER(i32)]),
    "rdm_profile_kind": (C.c_int, [i32, C.POINTER(C.c_char_p), C.POINTER(f64), C.POINTER(f64), C.POINTER(i32)]),
    "rdm_profile_kind_bytes": (f64, [i32]),
    "rdm_nyu_preprocess_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "rdm_nyu_preprocess": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "rdm_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "rdm_conv2d_dgrad": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "rdm_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "rdm_conv2d_fwd_ex": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "rdm_conv2d_fwd_bnsums": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, i32, vp]),
    "rdm_conv3x3_fwd_bnsums_acc": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, vp, i32, vp]),
    "rdm_conv2d_dgrad_ex": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp]),
    "rdm_conv2d_wgrad_ex": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp]),
    "rdm_conv2d_wgrad_x3": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, i32, vp]),
    "rdm_conv1x1_fwd_x6_workspace_bytes": (sz, [i32, i32]),
    "rdm_conv1x1_fwd_x6": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]),
    "rdm_conv1x1_dgrad_x3_workspace_bytes": (sz, [i32, i32]),
    "rdm_conv1x1_dgrad_x3": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, sz, i32, vp]),
    "rdm_conv3x3_dgrad_x3_workspace_bytes": (sz, [i32]),
    "rdm_conv3x3_dgrad_x3": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, sz, i32, vp]),
    "rdm_conv3x3_wino_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "rdm_conv3x3_wino_x6_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "rdm_conv3x3_wino_fwd_x6": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]),
    "rdm_conv3x3_wino_fwd": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]),
    "rdm_conv3x3_wino_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "rdm_conv3x3_wino_wgrad": (C.c_int, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, sz, vp]),
    "rdm_pack_conv_weight": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "rdm_unpack_conv_weight": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "rdm_gemm_bf16": (C.c_int, [vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "rdm_gemm_bf16_act": (C.c_int, [vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "rdm_conv3x3_bf16_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "rdm_conv3x3_bf16": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "rdm_conv3x3_act_bf16_weight_bytes": (sz, [i32]),
    "rdm_conv3x3_act_bf16_pack": (C.c_int, [vp, i32, vp, vp]),
    "rdm_conv3x3_act_bf16_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "rdm_conv3x3_act_bf16": (C.c_int, [vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "rdm_bn_stats": (C.c_int, [vp, i32, i64, i32, vp, vp, vp]),
    "rdm_bn_finalize": (C.c_int, [vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "rdm_bn_bwd_reduce": (C.c_int, [vp, i32, vp, i32, vp, vp, i64, i32, vp, vp, vp]),
    "rdm_bn_bwd": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, vp, f64, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "rdm_bn_bwd_defer": (C.c_int, [vp, i32, vp, i32, vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "rdm_maxpool3s2_fwd": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "rdm_maxpool3s2_bwd": (C.c_int, [vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "rdm_padavgpool2_fwd": (C.c_int, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "rdm_padavgpool2_bwd_workspace_bytes": (sz, [i32]),
    "rdm_padavgpool2_bwd": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "rdm_net_num_tensors": (C.c_int, []),
    "rdm_net_tensor_name": (C.c_char_p, [i32]),
    "rdm_net_tensor_numel": (i64, [i32]),
    "rdm_net_tensor_is_param": (C.c_int, [i32]),
    "rdm_net_create": (C.c_int, [i32, i32, i32, C.POINTER(vp)]),
    "rdm_net_destroy": (None, [vp]),
    "rdm_net_workspace_bytes": (sz, [vp]),
    "rdm_net_set_option": (C.c_int, [vp, i32, i32]),
    "rdm_net_output_hw": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "rdm_net_forward": (C.c_int, [vp, vp, C.POINTER(vp), vp, sz, vp, i32, vp]),
    "rdm_net_encoder_output": (C.c_int, [vp, vp, sz, vp, vp]),
    "rdm_net_bf16_weight_bytes": (sz, [vp]),
    "rdm_net_bf16_workspace_bytes": (sz, [vp]),
    "rdm_net_bf16_forward_bytes": (f64, [vp]),
    "rdm_net_bf16_prepare": (C.c_int, [vp, C.POINTER(vp), vp, sz, vp]),
    "rdm_net_forward_bf16": (C.c_int, [vp, vp, C.POINTER(vp), vp, sz, vp, sz, vp, vp]),
    "rdm_net_backward": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, sz, i32, i32, vp]),
    "rdm_net_segment_range": (C.c_int, [i32, C.POINTER(i32), C.POINTER(i32)]),
    "rdm_net_num_backward_stages": (C.c_int, []),
    "rdm_net_backward_stage_range": (C.c_int, [i32, C.POINTER(i32), C.POINTER(i32)]),
    "rdm_net_backward_stage": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, sz, i32, vp]),
    "rdm_net_buffer": (C.c_int, [vp, C.c_char_p, C.POINTER(i64), C.POINTER(i64)]),
    "rdm_net_forward_flops": (f64, [vp]),
    "rdm_net_backward_flops": (f64, [vp]),
    "rdm_dorn_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "rdm_dorn_bwd": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "rdm_ordinal_loss_fwd": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "rdm_ordinal_loss_bwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "rdm_depth2label_sid": (C.c_int, [vp, vp, i64, vp]),
    "rdm_depth2label_sid_ex": (C.c_int, [vp, vp, i64, i32, vp]),
    "rdm_resize_bicubic_f64": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "rdm_gm_normalize_f64": (C.c_int, [vp, vp, vp, i32, i32, f64, vp]),
    "rdm_decompose_f64": (C.c_int, [vp, vp, i32, i32, vp]),
    "rdm_fine_detail_pred_f32": (C.c_int, [vp, vp, vp, i32, i32, vp]),
    "rdm_fine_detail_pred_bwd": (C.c_int, [vp, vp, vp, i32, i32, vp]),
    "rdm_candidates_matvec_f32": (C.c_int, [vp, vp, vp, i32, i32, i64, vp]),
    "rdm_candidates_matvec_bwd": (C.c_int, [vp, vp, vp, i32, i32, i64, vp]),
    "rdm_split_rows_f32": (C.c_int, [vp, i32, vp, vp, vp, i32, i64, i32, vp]),
    "rdm_frame_split_rows_bytes": (sz, [i32, i32, i32]),
    "rdm_frame_split_rows_f32": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "rdm_layout_nchw_to_nhwc_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "rdm_layout_nhwc_to_nchw_f32": (C.c_int, [vp, i32, vp, i32, i32, i32, vp]),
    "rdm_recombine_f64": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "rdm_recombine_bwd": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "rdm_depth_metrics_f64": (C.c_int, [vp, vp, i64, vp, vp]),
    "rdm_ratio_grid_lloyd_dense": (C.c_int, [vp, vp, i32, i32, vp, vp, vp]),
    "rdm_ratio_grid_lloyd_paged": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, i32, vp]),
    "rdm_als_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "rdm_als_rank1": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "rdm_als_rank1_paged": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, i32, vp, sz, vp]),
    "rdm_page_split_f32": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "rdm_page_reconstruct_f32": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "rdm_adamw_fused": (C.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
}

# librdm_bench.so (include/rdm_bench.h): measurement kernels of tools/ and bench_ops.py - not part of the product
_BENCH_SIGNATURES = {
    "rdm_microbench_copy": (C.c_int, [vp, vp, i64, vp]),
    "rdm_microbench_mfma_f32": (C.c_int, [vp, i32, i32, vp]),
    "rdm_microbench_mfma_staged_f32": (C.c_int, [vp, i64, i32, i32, i32, i64, vp]),
    "rdm_microbench_gemm_dma_f32": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "rdm_microbench_xcd_sync": (C.c_int, [vp, vp, i32, i32, i32, vp, vp]),
}
BENCH_LIB_PATH = os.path.join(_HERE, "librdm_bench.so")

_lib = None
_bench = None


class RdmError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RdmError(f"{LIB_PATH} not found - run `python -m md_rdm_amd.build` (hipcc --offload-arch=gfx950). "
                           "There is no CPU/PyTorch fallback for the hot path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)     # AttributeError here = header/library drift: fail loudly
            fn.restype, fn.argtypes = res, args
        # development A/B switch (include/rdm_dev.h): exported by RDM_DEV_VARIANTS=1 builds only; on the shipped library the tools' calls
        # land in a Python stub that accepts 0 (= the shipped configuration) and refuses anything else
        try:
            dv = getattr(L, "rdm_debug_variant")
            dv.restype, dv.argtypes = None, [i32]
        except AttributeError:
            def _no_variants(v):
                if int(v) != 0:
                    raise RdmError(f"rdm_debug_variant({int(v)}): this library was built without RDM_DEV_VARIANTS - rebuild with RDM_DEV_VARIANTS=1 python -m md_rdm_amd.build")
            L.rdm_debug_variant = _no_variants
        _lib = L
    return _lib


def census():
    """{kernel variant name: launches} recorded since the last rdm_census_reset() (rdm_census_enable(1) must be on)."""
    L = lib()
    out = {}
    for i in range(L.rdm_census_count()):
        name, n = C.c_char_p(), i64()
        check(L.rdm_census_entry(i, C.byref(name), C.byref(n)))
        out[name.value.decode()] = n.value
    return out


def bench_lib():
    """librdm_bench.so, for development tools only (the product never loads it)."""
    global _bench
    if _bench is None:
        lib()                                       # the bench library resolves its support symbols from the product library
        if not os.path.exists(BENCH_LIB_PATH):
            raise RdmError(f"{BENCH_LIB_PATH} not found - run `python -m md_rdm_amd.build`")
        B = C.CDLL(BENCH_LIB_PATH)
        for name, (res, args) in _BENCH_SIGNATURES.items():
            fn = getattr(B, name)
            fn.restype, fn.argtypes = res, args
        _bench = B
    return _bench


def exported_symbols():
    return list(_SIGNATURES)


def bench_symbols():
    return list(_BENCH_SIGNATURES)


def check(rc):
    if rc != 0:
        raise RdmError(f"librdm_hip error {rc}: {lib().rdm_last_error_string().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RdmError("non-contiguous tensor passed to the C ABI")
    return C.c_void_p(t.data_ptr())


def stream():
    """The caller's current HIP stream.  Through the raw query where this torch has it: ``torch.cuda.current_stream()`` builds a Stream object and asks the
    runtime for the device count on the way (hipGetDeviceCount, ~20 us per call on this stack; ~100 C-ABI calls per training step; the host is two steps
    ahead of the GPU in steady state, so this only shortens the first step after a synchronisation - tools/host_pace.py)."""
    import torch

    raw, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
    if raw is not None and dev is not None:
        return C.c_void_p(raw(dev()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
