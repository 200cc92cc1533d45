#!/usr/bin/env python3
"""Benchmark of the CREStE perception->costmap hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): inference, batch = 16 synthetic frames per GPU of 1216x608 RGB +
a 128x1024 LiDAR range image projected to the sparse depth channel -> EfficientNet-B0 U-Net encoder ->
depth-guided BEV splat (256x256) -> ResNet-18 BEV heads -> reward FCN costmap (MaxEntIRL with
solve_mdp=False, the deployment graph of scripts/runtime/compile.py).  One "step" = one forward pass
over one batch; inputs are resident in HBM when the timed region starts; frames shard across GPUs as
independent replicas (no data-path collective): `value` = frames of all ranks / max-over-ranks time.

`value` is measured at the reference's arithmetic: fp32-equivalent operands (`bf16x6`: every fp32 operand as three
bf16 pieces = 24 significand bits, six piece products per multiply, fp32 accumulation); the narrower `f16x3` / `bf16`
modes are side entries of `modes_frames_per_s`.  The JSON line also carries `roofline` for the dominant kernel
(algorithmic FLOPs / HIP-event time per launch, summed over every launch of the timed steps), `roofline_splat` /
`roofline_vi` for the two HBM-bound kernels of the north star (HIP events around the BEV splat call inside the timed
steps; the value-iteration kernel alone), `value_host_fed` (the same steps with every batch copied from pinned host
memory inside the timed region, double-buffered on a copy stream) and `cpu_baseline` (the CPU oracle, i.e. the
reference's PyTorch op sequence, timed on this host).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # before the HIP runtime starts (see creste_public_amd/__init__.py)

IMG_H, IMG_W, BATCH = 608, 1216, 16
# TFLOP/s dense (MI355X_MICROARCH.md): the peak of the MFMA instruction each mode ISSUES -- v_mfma_f32_32x32x2_f32
# 157.3, v_mfma_f32_32x32x16_{bf16,f16} 2500.  `roofline.frac` = algorithmic conv FLOP/s / that peak; the split modes
# issue PRODUCTS[mode] MFMA products per algorithmic multiply, so the pipe's issue utilisation (`mfma_issue_util`) is
# PRODUCTS x frac.
PEAK = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0, "bf16x6": 2500.0, "f16x3": 2500.0,
        "bf16x6+winograd": 2500.0, "bf16x3+winograd": 2500.0, "bf16x6+winograd4": 2500.0, "bf16x3+winograd4": 2500.0}
# piece products issued per ALGORITHMIC multiply of the direct conv (Winograd F(2x2,3x3): 16 multiplies per 36)
PRODUCTS = {"f32": 1, "bf16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3, "bf16x6+winograd": 6 * 16 / 36, "bf16x3+winograd": 3 * 16 / 36,
            "bf16x6+winograd4": 6 * 36 / 144, "bf16x3+winograd4": 3 * 36 / 144}
PREC_NAME = {0: "f32", 1: "bf16", 2: "bf16x3", 3: "bf16x6", 4: "f16x3"}
DTYPE = {"f32": "f32 (exact fp32 products on v_mfma_f32_32x32x2_f32, fp32 accumulate)",
         "bf16x6": "bf16x6 (fp32 operands as 3 bf16 pieces = 24 significand bits, 6 piece products per multiply on the "
                   "bf16 MFMA, fp32 accumulate: fp32-equivalent products)",
         "f16x3": "f16x3 (fp32 operands rescaled by exact powers of two and split into fp16 hi+lo = 22 significand "
                  "bits, 3 piece products per multiply on the f16 MFMA, fp32 accumulate; product error <= 2^-21 -- "
                  "narrower than fp32 operands, inside the north-star tolerances: tests/test_fullsize_gpu.py)",
         "bf16x3": "bf16x3 (fp32 operands as 2 bf16 pieces = 16 significand bits, 3 piece products, fp32 accumulate)",
         "bf16": "bf16 operands, fp32 accumulate / activations"}
HBM_PEAK_GBS = 8000.0


def build_model(device):
    from creste_public_amd import MaxEntIRL, maxent_irl_cfg, synth
    torch.manual_seed(1337)
    model = MaxEntIRL(maxent_irl_cfg((IMG_H, IMG_W), solve_mdp=False))
    synth.randomize_bn(model, seed=1337)
    model = model.to(device).eval()
    # statistics of a trained network (see synth.calibrate_bn_hip): varied depths, ~21 % of the BEV cells occupied --
    # a random-init network would splat nothing and run the BEV heads on zeros
    rgbd, p2p = synth.make_frames(2, IMG_H, IMG_W, seed=4321)
    synth.calibrate_bn_hip(model, rgbd.to(device), p2p.to(device))
    return model


def kernel_symbol(pc, N, Ho, Wo):
    """The kernel a packed conv dispatches to (mirrors csrc/conv_igemm.hip / conv_patch.hip: patch_tn), named as
    rocprofv3 prints it, so bench numbers and profiles/ line up kernel by kernel."""
    if getattr(pc, "algo", 0) == 1:             # Winograd F(2x2,3x3): GEMM over the 16 transform positions + output transform
        split = {2: 2, 3: 3}[pc.prec]
        return (PREC_NAME[pc.prec] + "+winograd", f"wino_gemm_kernel<{split}, {4 if pc.Cout > 128 else 2}> + wino_out_kernel")
    if getattr(pc, "algo", 0) == 2:             # Winograd F(4x4,3x3): input transform + GEMM over the 36 positions + output transform
        split = {2: 2, 3: 3}[pc.prec]
        w4tn = 4 if pc.Cout > 128 else 2
        m_blocks = (N * ((Ho + 3) // 4) * ((Wo + 3) // 4) + 255) // 256
        if w4tn == 4 and m_blocks * 36 * ((pc.Cout + 255) // 256) <= 512:       # small maps: narrow tiles (conv_wino4_run)
            w4tn = 2
        return (PREC_NAME[pc.prec] + "+winograd4",
                f"wino4_in1_kernel<UP> + wino4_gemm32_kernel<{split}, {w4tn}, 0, true> + wino4_out2_kernel"
                " (conv3x3 -> conv3x3 pairs: wino4_outin_kernel in place of the first conv's output and the second conv's input transform)")
    if pc.prec == 0:
        return ("f32", "conv_igemm_f32_kernel<2, 2, 2, 2>" if pc.Cout > 64 else
                "conv_igemm_f32_kernel<2, 2, 2, 1>" if pc.Cout > 32 else "conv_igemm_f32_kernel<4, 1, 1, 1>")
    if pc.stride == 2 or pc.KH > 3:             # f16x3 only: row-at-a-time kernel
        tn2 = 2 if pc.Cout > 64 and not (pc.KH == 7 and pc.stride == 2) else 1
        if pc.prec == 3:                        # bf16x6: three pieces, 64-cout tiles (128 for the 1x1/2 downsamples)
            tn2 = 2 if (pc.KH == 1 and pc.Cout > 64) else 1
            return (PREC_NAME[pc.prec], f"conv_patch_row_kernel<{pc.KH}, {pc.stride}, {tn2}, 3, false>")
        return (PREC_NAME[pc.prec], f"conv_patch_row_kernel<{pc.KH}, {pc.stride}, {tn2}, 2, true>")
    split = {1: 1, 2: 2, 3: 3, 4: 2}[pc.prec]
    tn = (4 if (pc.prec == 4 or (pc.KH == 1 and pc.prec == 3)) else 2) if pc.Cout > 128 else 2 if pc.Cout > 64 else 1
    if pc.KH == 1 and pc.Cin < 256 and tn == 4:
        tn = 2
    px_tiles = N * ((Ho + 7) // 8) * ((Wo + 31) // 32)
    thr = (1600 if pc.Cin < 256 else 800) if pc.KH == 1 else 400
    while tn > 1 and px_tiles * ((pc.Cout + 64 * tn - 1) // (64 * tn)) < thr:
        tn //= 2
    f16 = "true" if pc.prec == 4 else "false"
    if pc.KH == 1 and pc.prec != 4 and (pc.Cin + 15) // 16 >= 4 and pc.pad_t == 0 and pc.pad_l == 0 and pc.stride == 1:
        # the deep-prefetch form on flat 128-pixel tiles (conv_patch_run): every such conv with > 64 output channels that is
        # not on the 256-wide tile, and the narrow ones while <= 1024 workgroups of 256 pixels
        pt = (N * Ho * Wo + 255) // 256
        n128, n64 = pt * ((pc.Cout + 127) // 128), pt * ((pc.Cout + 63) // 64)
        if (tn <= 2) if pc.Cout > 64 else n64 <= 1024:
            dtn = 2 if (pc.Cout > 64 and n128 >= 1024) else 1
            return (PREC_NAME[pc.prec], f"conv1x1_deep_kernel<{split}, {dtn}, 1, GATED>")
    name = (f"conv_patch3_kernel<{split}, {tn}, {f16}, false>" if pc.KH == 3
            else f"conv_patch_kernel<1, {split}, {tn}, {f16}>")
    return (PREC_NAME[pc.prec], name)


class ConvProfiler:
    """HIP-event pair around every conv launch (recorded on the stream the kernel is launched on)."""

    def __init__(self):
        self.records = []
        self.splat = []          # (event0, event1, algorithmic bytes) of every BEV splat call

    def install(self):
        from creste_public_amd import ops
        self._ops, self._orig = ops, ops.conv2d
        self._orig_plan, self._orig_gather = ops.bev_splat_plan, ops.bev_splat_gather
        self._orig_plan_keyed = ops.bev_splat_plan_keyed
        prof = self

        def timed_plan_keyed(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan = prof._orig_plan_keyed(*a, **kw)
            e1.record()
            plan._bench_ev = (e0, e1)
            return plan
        ops.bev_splat_plan_keyed = timed_plan_keyed

        def timed_plan(xyz, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            plan = prof._orig_plan(xyz, *a, **kw)
            e1.record()
            plan._bench_ev = (e0, e1)
            return plan

        def timed_gather(plan, feats, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = prof._orig_gather(plan, feats, *a, **kw)
            e1.record()
            B, P, F, GH, GW = plan.B, plan.P, feats.C, plan.GH, plan.GW
            # 4 * (F*P + 2*P + F*G + G) bytes per frame -- features and xy read once, BEV map + densities written once
            prof.splat.append((plan._bench_ev, (e0, e1), 4.0 * B * (F * P + 2 * P + F * GH * GW + GH * GW)))
            return r
        ops.bev_splat_plan, ops.bev_splat_gather = timed_plan, timed_gather

        def timed(x, pc, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = prof._orig(x, pc, **kw)
            e1.record()
            flops = 2.0 * y.N * y.H * y.W * pc.Cout * pc.Cin * pc.KH * pc.KW
            prof.records.append((e0, e1, flops, kernel_symbol(pc, y.N, y.H, y.W), (pc.Cin, pc.Cout, pc.KH, y.H, y.W)))
            return y
        import creste_public_amd.hipnn as hipnn
        ops.conv2d = timed
        hipnn.ops.conv2d = timed
        self._orig_up, self._orig_chain = ops.upconv2x, ops.conv1x1_chain3

        def timed_up(x, pu, **kw):
            # Upsample(x2) -> conv3x3 as phase convolutions: one F(4x4,3x3) call of the same kernels (+ the ring fix); FLOPs = the
            # reference operator's, a direct 3x3 conv over the UPSAMPLED map
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = prof._orig_up(x, pu, **kw)
            e1.record()
            flops = 2.0 * y.N * y.H * y.W * pu.Cout * pu.phase.Cin * 9
            prof.records.append((e0, e1, flops, kernel_symbol(pu.phase, x.N, x.H, x.W), (pu.phase.Cin, pu.Cout, 3, y.H, y.W)))
            return y

        def timed_chain(x, pk, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = prof._orig_chain(x, pk, **kw)
            e1.record()
            flops = 2.0 * y.N * y.H * y.W * 128 * (pk.Cin + 256)
            prof.records.append((e0, e1, flops, (PREC_NAME[pk.prec], f"conv1x1_chain3_kernel<{3 if pk.prec == 3 else 2}>"),
                                 (pk.Cin, 128, 1, y.H, y.W)))
            return y
        ops.upconv2x, ops.conv1x1_chain3 = timed_up, timed_chain

    def uninstall(self):
        self._ops.conv2d = self._orig
        self._ops.upconv2x, self._ops.conv1x1_chain3 = self._orig_up, self._orig_chain
        self._ops.bev_splat_plan, self._ops.bev_splat_gather = self._orig_plan, self._orig_gather
        self._ops.bev_splat_plan_keyed = self._orig_plan_keyed

    def splat_roofline(self, key_ms: float = 0.0):
        """key_ms: what the plan's first kernel (voxel coordinates + base-cell keys) adds to the pixel-geometry kernel it now
        runs inside (keyed_geometry_extra_ms below), charged to the plan"""
        if not self.splat:
            return None
        plan_ms = [p[0].elapsed_time(p[1]) + key_ms for p, _, _ in self.splat]
        gath_ms = [g[0].elapsed_time(g[1]) for _, g, _ in self.splat]
        ms = [a + b for a, b in zip(plan_ms, gath_ms)]
        by = self.splat[0][2]
        avg = sum(ms) / len(ms)
        return {"bound": "hbm", "kernel": "the key computation inside pixel_geometry_px_kernel<true> (its measured extra time) + "
                                          "creste_bev_splat_plan_keyed_f32 (splat_build_reg + splat_fill_rec + splat_sort_rec) + "
                                          "creste_bev_splat_gather_f32 (splat_gather8): every splat kernel of the step",
                "key_in_geometry_ms": round(key_ms, 4),
                "achieved": round(by / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(by / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes": by, "avg_call_ms": round(avg, 4),
                "plan_ms": round(sum(plan_ms) / len(plan_ms), 4), "gather_ms": round(sum(gath_ms) / len(gath_ms), 4),
                "min_call_ms": round(min(ms), 4), "calls": len(ms), "traffic": None,
                "note": "HIP events around the binning plan (enqueued right after the pixel geometry, ahead of the fusion "
                        "conv) and around the gather, in the one-stream steps run right AFTER the timed region, not inside it (the "
                        "network's own predicted depths and fused features, batch 16); time = plan_ms + gather_ms; bytes = 4*(F*P + 2*P + F*G + G) per frame "
                        "(SURVEY 8d: F=96, P=46208, G=65536), one read of the inputs and one write of the outputs"}

    def summary(self):
        by = {}
        for e0, e1, fl, bn, shape in self.records:
            ms = e0.elapsed_time(e1)
            d = by.setdefault(bn, dict(ms=0.0, flops=0.0, n=0))
            d["ms"] += ms
            d["flops"] += fl
            d["n"] += 1
        return by


def gemm_kernel_probe(step):
    """One extra (untimed) step with HIP events around the GEMM KERNEL of every F(4x4,3x3) conv call (library probe:
    creste_conv_wino4_gemm_probe): the matrix-core kernel alone, without its two bandwidth-bound transform kernels."""
    import ctypes as C
    from creste_public_amd import ops, _lib
    import creste_public_amd.hipnn as hipnn
    lib = _lib.load()
    orig = ops.conv2d
    ms_tot, raw, alg, calls = 0.0, 0.0, 0.0, 0

    def probed(x, pc, **kw):
        nonlocal ms_tot, raw, alg, calls
        if getattr(pc, "algo", 0) != 2:
            return orig(x, pc, **kw)
        lib.creste_conv_wino4_gemm_probe(1)
        y = orig(x, pc, **kw)
        lib.creste_conv_wino4_gemm_probe(0)
        ms = C.c_float(0.0)
        _lib.check(lib.creste_conv_wino4_gemm_last_ms(C.byref(ms)), "gemm_last_ms")
        T = y.N * ((y.H + 3) // 4) * ((y.W + 3) // 4)
        tn = 256 if pc.Cout > 128 else 128
        mp, kp, np_ = (T + 255) // 256 * 256, (pc.Cin + 15) // 16 * 16, (pc.Cout + tn - 1) // tn * tn
        split = {2: 3, 3: 6}[pc.prec]                     # piece products per multiply
        ms_tot += ms.value
        raw += 36.0 * mp * kp * np_ * 2.0 * split         # issued, padded tiles included
        alg += 2.0 * y.N * y.H * y.W * pc.Cout * pc.Cin * 9
        calls += 1
        return y
    orig_up = ops.upconv2x

    def probed_up(x, pu, **kw):
        nonlocal ms_tot, raw, alg, calls
        lib.creste_conv_wino4_gemm_probe(1)
        y = orig_up(x, pu, **kw)
        lib.creste_conv_wino4_gemm_probe(0)
        ms = C.c_float(0.0)
        _lib.check(lib.creste_conv_wino4_gemm_last_ms(C.byref(ms)), "gemm_last_ms")
        pc = pu.phase
        T = x.N * ((x.H + 3) // 4) * ((x.W + 3) // 4)
        mp, kp, np_ = (T + 255) // 256 * 256, (pc.Cin + 15) // 16 * 16, (pc.Cout + 255) // 256 * 256
        ms_tot += ms.value
        raw += 36.0 * mp * kp * np_ * 2.0 * {2: 3, 3: 6}[pc.prec]
        alg += 2.0 * y.N * y.H * y.W * pu.Cout * pc.Cin * 9
        calls += 1
        return y
    ops.conv2d = probed
    hipnn.ops.conv2d = probed
    ops.upconv2x = probed_up
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        hipnn.ops.conv2d = orig
        ops.upconv2x = orig_up
    if not calls:
        return None
    return {"kernel": "wino4_gemm32_kernel<SPLIT, TN> (the GEMM kernel of every F(4x4,3x3) conv call of one step)",
            "calls_per_step": calls, "ms_per_step": round(ms_tot, 3),
            "piece_products_pflops": round(raw / (ms_tot * 1e-3) / 1e15, 4),
            "mfma_issue_util": round(raw / (ms_tot * 1e-3) / 1e12 / 2500.0, 4),
            "algorithmic_tflops": round(alg / (ms_tot * 1e-3) / 1e12, 1),
            "note": "HIP events around the GEMM kernel alone (library probe), one untimed step after the timed ones; "
                    "piece products = 36 positions x padded tiles x padded Cin x padded Cout x 2 x pieces, issued on "
                    "v_mfma_f32_32x32x16_bf16 (dense peak 2500 TFLOP/s); SQ_VALU_MFMA_BUSY_CYCLES of this kernel: roofline.mfma_busy_gemm (profiles/r06_pmc_encoder.txt)"}


def cpu_baseline(batches=(1, 16), runs=5, budget_s=210.0):
    """The oracle (CPU restatement of the reference's PyTorch path) on this host's cores: batch 1 and batch 16,
    median of `runs` timed runs after 1 warm-up each (SURVEY 8d / BASELINE.md section 2).  `value` is the better of the
    two batch sizes in frames/s.  `budget_s` bounds the batch-16 leg: if its warm-up shows that `runs` runs would not
    fit, fewer runs are timed and the count is reported."""
    from creste_public_amd import maxent_irl_cfg, synth
    from oracle.irl import MaxEntIRL as OracleIRL
    from oracle import lidar as olidar
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    torch.set_num_threads(cores)
    torch.manual_seed(1337)
    model = OracleIRL(maxent_irl_cfg((IMG_H, IMG_W), solve_mdp=False)).eval()
    blas = " ".join(l.strip() for l in torch.__config__.show().splitlines()
                    if any(k in l for k in ("MKL", "oneDNN", "BLAS", "OpenMP", "LAPACK")) and "flags" not in l)[:300]
    per = {}
    for nb in batches:
        gen = torch.Generator().manual_seed(1337)
        rgbd = torch.zeros(nb, 1, 4, IMG_H, IMG_W)
        rgbd[:, 0, :3] = torch.rand(nb, 3, IMG_H, IMG_W, generator=gen)
        scan = synth.lidar_scan(nb, gen).numpy()
        l2c = synth.lidar2camrect(nb, IMG_H, IMG_W).numpy()
        p2p = synth.make_p2p(nb, IMG_H, IMG_W)

        def cpu_step():
            for b in range(nb):     # LiDAR scan -> sparse mm depth channel, then the forward
                rgbd[b, 0, 3] = torch.from_numpy(olidar.depth_image(scan[b], l2c[b], IMG_H, IMG_W) * 1000.0).float()
            return model((rgbd, p2p))

        with torch.no_grad():
            t0 = time.perf_counter()
            cpu_step()                               # warm-up
            warm = time.perf_counter() - t0
            n = runs if nb == 1 else max(1, min(runs, int(budget_s / max(warm, 1e-3))))
            times = []
            for _ in range(n):
                t0 = time.perf_counter()
                cpu_step()
                times.append(time.perf_counter() - t0)
        per[nb] = {"frames_per_s": round(nb / statistics.median(times), 4), "runs": n,
                   "median_s": round(statistics.median(times), 3)}
    best = max(per, key=lambda k: per[k]["frames_per_s"])
    return {"value": per[best]["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU PyTorch restatement of the reference path, fp32, eval, no_grad), {IMG_W}x{IMG_H} "
                      f"frames incl. the LiDAR projection, {cores} threads (torch.set_num_threads), "
                      + "; ".join(f"batch {nb}: {v['frames_per_s']} frames/s (median of {v['runs']} runs after 1 warm-up, "
                                  f"{v['median_s']} s/run)" for nb, v in per.items())
                      + f"; value = batch {best}",
            "by_batch": {str(k): v for k, v in per.items()}, "blas": blas}


def _median_step_ms(fn, steps):
    """median wall time of `steps` synchronised calls (a single hiccup -- another tenant of the box, a stalled copy --
    tripled a 3-step mean once: the side measurements report the median)"""
    ts, last = [], None
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], last


def _distill_setup(device, B=8, seed=0):
    """-> (step_fn, describe_fn, trainer): one stage-1 distillation training step on `B` frames (see distill_extras)."""
    from creste_public_amd import harness, synth
    from creste_public_amd.creste.models.distillation import DistillationBackbone
    from creste_public_amd.creste.utils.loss_utils import LossManager
    cfg = harness.distillation_cfg((IMG_H, IMG_W))
    torch.cuda.empty_cache()                   # the inference / IRL runs before this leave a fragmented cache
    torch.manual_seed(0)                       # identical replicas on every rank; the DATA differs by `seed`
    model = DistillationBackbone(cfg).to(device)
    synth.randomize_bn(model, seed=1)
    rgbd, _ = synth.make_frames(B, IMG_H, IMG_W, seed=2 + 100 * seed)
    g = torch.Generator().manual_seed(3 + 100 * seed)
    batch = {"image": rgbd.to(device),
             "depth_label": (torch.rand(B, 1, IMG_H // 4, IMG_W // 4, generator=g) * 26000.0).to(device),
             "fimg_label": torch.randn(B, 1, 128, IMG_H // 4, IMG_W // 4, generator=g).to(device)}
    tr = harness.DistillTrainer(model, LossManager(cfg), cfg)

    def describe(logs):
        n = sum(p.numel() for p in model.parameters() if p.grad is not None)
        return (f"batch {B}, {IMG_W}x{IMG_H}, EfficientNet-B0 U-Net + depth/DINO heads in training mode, "
                f"{n} parameters with gradients, Adam; loss {float(logs['train/loss']):.3f}")
    return (lambda: tr.training_step(batch)), describe, tr


def distill_extras(device, steps=5, B=8):
    """BASELINE configs[3], per-GPU part: one stage-1 distillation training step (train_pefree.py) -- training-mode
    forward, CrossEntropyDepth + SmoothL1Depth + MSELoss, backward to all 25.5 M encoder parameters into the flat
    all-reduce arena, Adam -- batch 8 of 1216x608 on the HIP training kernels."""
    step, describe, tr = _distill_setup(device, B)
    step(); step(); torch.cuda.synchronize()
    ms, logs = _median_step_ms(step, steps)
    cfg_s = describe(logs)
    del tr, step, describe
    torch.cuda.empty_cache()
    import creste_public_amd as _cpa
    return {"distill_train_step_ms": round(ms, 1), "distill_frames_per_s": round(B / ms * 1e3, 1),
            "operands": _cpa.get_precision(), "distill_config": cfg_s}


def _ssc_setup(device, B=8, seed=0):
    """-> (step_fn, describe_fn, trainer): one BEV-SSC training step on `B` frames (see ssc_extras)."""
    from creste_public_amd import harness, synth
    from creste_public_amd.creste.models.terrainnet import TerrainNet
    from creste_public_amd.creste.utils.loss_utils import LossManager
    cfg = harness.ssc_cfg((IMG_H, IMG_W), class_weights=[0.5, 0.2, 0.1, 0.1, 0.05, 0.05])
    torch.manual_seed(0)
    model = TerrainNet(cfg).to(device)
    synth.randomize_bn(model, seed=1)
    synth.peak_depth_head(model)                 # varied depths -> a populated BEV map (see synth.calibrate_bn_hip)
    rgbd, p2p = synth.make_frames(B, IMG_H, IMG_W, seed=2 + 100 * seed)
    g = torch.Generator().manual_seed(3 + 100 * seed)
    G, Hs, Ws = 256, IMG_H // 4, IMG_W // 4
    data = {"image": rgbd, "p2p": p2p, "depth_label": torch.rand(B, 1, Hs, Ws, generator=g) * 26000.0,
            "fimg_label": torch.randn(B, 1, 128, Hs, Ws, generator=g),
            "3d_sam_label": torch.randint(0, 5, (B, 1, G // 16, G // 16), generator=g).repeat_interleave(16, 2)
            .repeat_interleave(16, 3),
            "3d_sam_dynamic_label": torch.stack([torch.zeros(B, G, G), torch.randint(0, 6, (B, G // 8, G // 8), generator=g)
                                                 .float().repeat_interleave(8, 1).repeat_interleave(8, 2)], dim=1),
            "fov_mask": torch.rand(B, G, G, generator=g) > 0.5, "elevation_label": torch.randn(B, 2, G, G, generator=g)}
    batch = {"joint": {k: v.to(device) for k, v in data.items()}}
    tr = harness.SSCTrainer(model, LossManager(cfg).to(device), cfg)

    def describe(logs):
        n = sum(p.numel() for p in model.parameters() if p.grad is not None)
        return (f"batch {B}, {IMG_W}x{IMG_H} -> 256x256 BEV, TerrainNet in training mode, SupPixelCon + CE + MSE + "
                f"depth CE + depth SmoothL1 + elevation SmoothL1, {n} parameters with gradients, Adam; "
                f"loss {float(logs['train/loss']):.3f}")
    return (lambda: tr.training_step(batch)), describe, tr


def ssc_extras(device, steps=5, B=8):
    """BASELINE configs[3], second stage: one BEV-SSC training step (train_ssc.py) -- TerrainNet in training mode
    (encoder + depth-guided splat + ResNet-18 BEV heads), the six SSC losses, backward through the splat into
    features and depth, Adam -- batch 8 of 1216x608 -> 256x256 BEV on the HIP training kernels."""
    step, describe, tr = _ssc_setup(device, B)
    step(); step(); torch.cuda.synchronize()
    ms, logs = _median_step_ms(step, steps)
    cfg_s = describe(logs)
    del tr, step, describe
    torch.cuda.empty_cache()
    import creste_public_amd as _cpa
    return {"ssc_train_step_ms": round(ms, 1), "ssc_frames_per_s": round(B / ms * 1e3, 1),
            "operands": _cpa.get_precision(), "ssc_config": cfg_s}


IRL_VARIANTS = {
    # reference config: 256x256 BEV (0.1 m voxels over +-12.8 m), map_ds 2 + front-half crop -> 64x128 MDP grid
    "reference": dict(pcr=None, voxel=None, map_size=(64, 128), map_ds=2, bev=(256, 256), prec=None, B=8),
    # BASELINE configs[2]: the MDP solved on a 256x256 grid.  The reference's geometry (max-pool by map_ds, front-half
    # crop, vin.py:104-109) gives an (R/ds/2) x (C/ds) grid from an R x C BEV map: 256x256 = the front half of a
    # 512x256 BEV map (0.1 m voxels, 51.2 m ahead/behind x 25.6 m across) with map_ds 1
    "mdp256": dict(pcr=[-25.6, -12.8, -2, 25.6, 12.8, 1], voxel=None, map_size=(256, 256), map_ds=1, bev=(512, 256),
                   prec=None, B=8),
    # BASELINE configs[4], per-GPU part: counterfactual IRL on a 512x512 BEV grid (5 cm voxels) -> 128x256 MDP grid,
    # bf16 encoder operands, fp32 reward network / value iteration / SVF
    "cf512": dict(pcr=None, voxel=[0.05, 0.05, 3], map_size=(128, 256), map_ds=2, bev=(512, 512), prec="bf16", B=8),
}


def irl_step_bench(model_infer, device, variant, steps=5):
    """One IRL training step (reference train_traversability.py:66-105) through harness.IRLTrainer -- the product path: frozen HIP
    backbone forward, reward net (train mode, hipGraph-replayed HIP kernels), value iteration + expected SVF, MaxEntIRLLoss with
    counterfactual mixing (alpha 0.5) and gradient penalty, backward incl. the second-order term, Adam; with the look-ahead batch
    (`training_step(batch, next_batch)`: the next frozen half on a side stream, enqueued ahead of this step's reward forward and solve
    since round 6; an aborted solve is redone in the launch-per-chunk form) and back to back as the reference."""
    import creste_public_amd
    v = IRL_VARIANTS[variant]
    B, (GH, GW) = v["B"], v["bev"]
    prev = creste_public_amd.get_precision()
    if v["prec"]:
        creste_public_amd.set_precision(v["prec"])
    try:
        step, tr = _irl_setup(model_infer, device, variant)
        batch = next(c.cell_contents for c in step.__closure__ if isinstance(c.cell_contents, dict))
        serial = lambda: tr.training_step(batch, None)
        def steady(fn, n):
            """n steps back to back, ONE synchronisation at the end (a training loop does not wait for the device after every
            step): wall time / n, the better of two repeats"""
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) / n * 1e3
                best = ms1 if best is None else min(best, ms1)
            return best
        n_steady = max(2 * steps, 10)
        serial(); serial(); torch.cuda.synchronize()          # eager step, then the capturing step
        ms_serial_sync, _ = _median_step_ms(serial, steps)
        ms_serial = steady(serial, n_steady)
        step(); step(); torch.cuda.synchronize()
        ms_sync, logs = _median_step_ms(step, steps)
        ms = steady(step, n_steady)
        tr.model._prefetched = None
        creste_public_amd.ops.vi_check(wait=True)
        sweeps = int(tr.model.traversability_head.last_sweeps.item())
        with torch.no_grad():
            d = batch["irl"]
            occ = float((tr.model((d["image"], d["p2p"], d["traversability_label"]))["bev_densities"] > 0).float().mean())
        creste_public_amd.ops.vi_check(wait=True)
        res = {"train_step_ms": round(ms, 2), "train_step_serial_ms": round(ms_serial, 2),
               "synced": {"train_step_ms": round(ms_sync, 2), "train_step_serial_ms": round(ms_serial_sync, 2),
                          "note": "median of steps each bracketed by torch.cuda.synchronize() (the method of the r01-r05 lines)"},
               "batch": B, "bev_grid": [GH, GW], "mdp_grid": list(v["map_size"]),
               "vi_sweeps": sweeps, "vi_retries": int(tr.vi_retries), "encoder_operands": creste_public_amd.get_precision(),
               "loss": round(float(logs["train/loss"]), 5), "bev_cells_occupied": None if occ is None else round(occ, 3)}
        del tr, step, serial, batch
        torch.cuda.empty_cache()
        return res
    finally:
        creste_public_amd.set_precision(prev)


def _irl_setup(model_infer, device, variant, seed=0):
    """-> (step_fn, trainer): one IRL training step of `IRL_VARIANTS[variant]` through harness.IRLTrainer (manual
    optimisation, look-ahead batch on the side stream, ONE flat gradient all-reduce: train_traversability.py:66-105)."""
    import numpy as np
    from creste_public_amd import LossManager, MaxEntIRL, harness, maxent_irl_cfg, synth
    v = IRL_VARIANTS[variant]
    B, (GH, GW) = v["B"], v["bev"]
    cfg = maxent_irl_cfg((IMG_H, IMG_W), solve_mdp=True, map_size=v["map_size"], map_ds=v["map_ds"],
                         point_cloud_range=v["pcr"], voxel_size=v["voxel"])
    model = MaxEntIRL(cfg)
    sd = {k: t for k, t in model_infer.state_dict().items() if ".cam2map." not in k or "z_proj" in k or "vision_fusion" in k}
    model.load_state_dict(sd, strict=False)
    with torch.no_grad():
        model.traversability_head.r.postpool[0].norm.weight.mul_(0.01)
        model.traversability_head.r.postpool[0].norm.bias.mul_(0.01)
    model = model.to(device).train()
    tr = harness.IRLTrainer(model, LossManager(cfg).to(device), cfg, graphs=True)
    rgbd, p2p = synth.make_frames(B, IMG_H, IMG_W, seed=4242 + 100 * seed)
    fov = torch.ones(B, max(GH, 2 * v["map_size"][0]), max(GW, 2 * v["map_size"][1]), dtype=torch.bool, device=device)
    rng = np.random.RandomState(seed)
    c0 = np.array([[GH / 2 - 28.0, GW / 2.0]])
    cf = [dict(trajectories=(c0 + np.linspace(0, 1, 20)[None, :, None] *
                             rng.uniform(-0.3 * GW, 0.3 * GW, size=(2, 1, 2))).astype(np.float32),
               rank=np.array([0, 1])) for _ in range(B)]
    batch = {"irl": {"image": rgbd.to(device), "p2p": p2p.to(device),
                     "traversability_label": synth.make_experts(B, 50, (GH, GW), seed=5 + seed).to(device),
                     "fov_mask": fov, "counterfactuals_label": cf}}
    return (lambda: tr.training_step(batch, batch)), tr


def train_dp_extras(model_infer, device, rank, steps=5):
    """BASELINE configs[3] / [4]: the data-parallel training steps under the process group bench.py was launched with --
    every rank a micro-batch of 8 frames of its own, synchronous SGD through harness.DistillTrainer / SSCTrainer /
    IRLTrainer with the real gradient exchange (dist_utils.GradArena inside the encoder's backward, HookedArena on the
    BEV heads, one flat all-reduce of the reward net; reference: Lightning DDP, train_pefree.py:261-288,
    train_ssc.py:342-358, train_traversability.py:400-416).  Per leg: dist_utils.measure_dp_step."""
    import creste_public_amd
    from creste_public_amd import dist_utils
    out = {"note": "per-rank micro-batch 8 x 1216x608; step_ms = max over ranks of the median step; allreduce_exposed_ms = "
                   "step_ms - the same steps with the gradient collectives switched off (what the overlap did not hide); "
                   "allreduce_bytes / collective_calls = payload this rank hands to RCCL per step; frames_per_s = whole job"}

    def leg(name, setup):
        torch.cuda.empty_cache()
        objs = setup()
        step = objs[0]
        step(); torch.cuda.synchronize()                      # eager / graph-capturing first step
        out[name] = dist_utils.measure_dp_step(step, steps, 8, device=device, warmup=1)
        out[name]["operands"] = creste_public_amd.get_precision()
        tr = objs[-1]
        if getattr(tr, "model", None) is not None and hasattr(tr.model, "_prefetched"):
            tr.model._prefetched = None
        del objs, step, tr
        torch.cuda.empty_cache()

    leg("distill", lambda: _distill_setup(device, 8, seed=rank))
    leg("ssc", lambda: _ssc_setup(device, 8, seed=rank))
    leg("irl_reference", lambda: _irl_setup(model_infer, device, "reference", seed=rank))
    prev = creste_public_amd.get_precision()
    creste_public_amd.set_precision("bf16")                   # configs[4]: bf16 encoder + fp32 IRL sweep
    try:
        leg("irl_cf512", lambda: _irl_setup(model_infer, device, "cf512", seed=rank))
    finally:
        creste_public_amd.set_precision(prev)
    return out


def vi_kernel_bench(device, B, Hg, Wg):
    """The MDP kernels alone on r ~ U[0,1): value iteration (gamma 0.99, threshold 1e-3)."""
    from creste_public_amd import ops
    r = torch.rand(B, Hg, Wg, device=device)
    ops.value_iteration(r, 0.99, 1e-3); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        v, q, pi, sw = ops.value_iteration(r, 0.99, 1e-3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    n = int(sw.item())
    gbs = B * Hg * Wg * (12 * n + 72) / ms / 1e6
    return {"ms": round(ms, 3), "sweeps": n, "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3)}


def irl_extras(model_infer, device, steps=5):
    """The second half of BASELINE.json's metric: IRL train-step time, at the reference config, at BASELINE configs[2]
    (256x256 MDP grid) and at configs[4]'s per-GPU shape (512x512 BEV, bf16 encoder, counterfactual IRL)."""
    out = {"config": "frames 1216x608; frozen HIP backbone + reward net training kernels + VI + SVF (T=50) + CF-IRL loss "
                     "(alpha 0.5, 2 counterfactual trajectories per sample) + gradient penalty + Adam; train_step_ms: the "
                     "next batch's frozen-backbone forward runs on a side stream under this batch's whole trainable half, reward forward "
                     "and MDP solve included (harness.IRLTrainer.training_step(batch, next_batch); vi_retries = solves redone in the "
                     "launch-per-chunk form after an abort); train_step_serial_ms: back to back as the reference.  Both are STEADY-STATE step "
                     "times: 2 x steps training steps enqueued back to back, one synchronisation at the end, wall time / steps (a training "
                     "loop does not wait for the device after every step; with a synchronisation per step the next step's host-side "
                     "launches cannot run ahead, which costs the pipelined step ~1.4 ms); `synced` = the r01-r05 method for comparison"}
    for name in IRL_VARIANTS:
        out[name] = irl_step_bench(model_infer, device, name, steps)
    out["irl_train_step_ms"] = out["reference"]["train_step_ms"]
    out["vi_8x64x128"] = vi_kernel_bench(device, 8, 64, 128)
    out["vi_8x256x256"] = vi_kernel_bench(device, 8, 256, 256)
    out["vi_8x512x512"] = vi_kernel_bench(device, 8, 512, 512)
    return out


def keyed_geometry_extra_ms(device, reps=20) -> float:
    """The splat plan's key kernel runs inside the pixel-geometry kernel: its cost = keyed launch - plain launch of the same
    geometry (batch 16, the bench's feature-map size), HIP events around `reps` launches each, best of 3; >= 0."""
    from creste_public_amd import ops
    Hs, Ws, B = IMG_H // 4, IMG_W // 4, BATCH
    g = torch.Generator().manual_seed(5)
    depth = (torch.rand(B, Hs, Ws, generator=g) * 20 + 1).to(device)
    p2p = torch.eye(4).repeat(B, 1, 1).to(device)
    bounds = torch.tensor([-12.8, -12.8, -2.0, 12.8, 12.8, 1.0], device=device)
    w1, b1 = torch.randn(64, generator=g).to(device), torch.randn(64, generator=g).to(device)
    w2, b2 = torch.randn(32, 64, generator=g).to(device), torch.randn(32, generator=g).to(device)
    z = ops.Act.empty(B, Hs, Ws, 32, device)
    lib = ops._lib.load()
    xyz = torch.empty(B, Hs * Ws, 3, device=device); mask = torch.empty(B, Hs * Ws, device=device)
    coords = torch.empty(B, Hs * Ws, 2, device=device)
    work = torch.empty(lib.creste_bev_splat_workspace_bytes(B, Hs * Ws, 256, 256), dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream
    common = (depth.data_ptr(), p2p.data_ptr(), B, Hs, Ws, bounds.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
              b2.data_ptr(), 64, 32, xyz.data_ptr(), mask.data_ptr(), z.buf.data_ptr(), z.cs, z.co)

    def plain():
        ops._lib.check(lib.creste_pixel_geometry_f32(*common, st), "pixel_geometry")

    def keyed():
        ops._lib.check(lib.creste_pixel_geometry_keyed_f32(*common, 12.8, 12.8, 0.1, 0.1, 256, 256, coords.data_ptr(),
                                                           work.data_ptr(), st), "pixel_geometry_keyed")

    def t(fn):
        best = 1e9
        for _ in range(3):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best
    return max(0.0, t(keyed) - t(plain))


def latency_extras(model, device, iters=30):
    """single-frame latency of the same forward (the deployed robot runs batch 1: scripts/runtime in the reference)"""
    from creste_public_amd import synth
    rgbd, p2p = synth.make_frames(1, IMG_H, IMG_W, seed=7)
    rgbd, p2p = rgbd.to(device), p2p.to(device)
    with torch.no_grad():
        for _ in range(3):
            model((rgbd, p2p))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model((rgbd, p2p))
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"batch": 1, "ms_per_frame": round(ms, 3), "frames_per_s": round(1e3 / ms, 1),
            "note": "eager launches through the C ABI, inputs resident in HBM; not the headline workload"}


def self_launch(n: int, argv=None) -> int:
    """Re-run this script as `n` ranks: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` (what the driver runs for N > 1; CRESTE_BENCH_FORCE_LAUNCH=1 takes this
    path for N = 1 too).  The children inherit stdout / stderr: rank 0's JSON line is this command's JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CRESTE_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))     # (the launcher's default of 1 would
    #                                                                                throttle the rank-0 CPU baseline)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    # The parent is the outer watchdog: it passes the children's stdout through, notes whether rank 0's JSON line went by, and
    # when the launcher dies / a rank fails (torch.distributed.run then tears the others down) / CRESTE_BENCH_TIMEOUT_S passes
    # (default 1700 s, below the driver's 1800 s limit) without one, prints a line with an `error` field itself instead of
    # leaving the caller with nothing
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    seen = {"line": False}

    def pump():
        for ln in proc.stdout:
            if ln.lstrip().startswith("{") and '"metric"' in ln:
                seen["line"] = True
            sys.stdout.write(ln)
            sys.stdout.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    limit = float(os.environ.get("CRESTE_BENCH_TIMEOUT_S", "1700"))
    err = None
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        err = f"no result within {limit:.0f} s: the {n}-rank job was killed (a rank hung?)"
        try:
            os.killpg(proc.pid, 15)                  # our own process group (start_new_session): launcher + every rank
            proc.wait(timeout=20)
        except Exception:
            try:
                os.killpg(proc.pid, 9)
            except Exception:
                pass
        rc = 124
    th.join(timeout=10)
    if not seen["line"]:
        print(json.dumps(error_line(n, err or f"torch.distributed.run exited with code {rc} before rank 0 printed its line "
                                               f"(a rank failed; see stderr)")), flush=True)
        rc = rc or 1
    return rc


def error_line(n_gpus: int, message: str, **extra) -> dict:
    """the ONE JSON line of a run that could not produce a measurement: same leading keys, value null, an `error` field"""
    line = {"metric": "frames/sec (RGB+LiDAR->BEV costmap)", "value": None, "unit": "frames/s", "n_gpus": n_gpus,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "error": message}
    line.update(extra)
    return line


class RankWatchdog:
    """In-rank half of the watchdog (the driver starts N > 1 ranks under ITS OWN torch.distributed.run, no parent of ours):
    every collective / barrier of the bench is bracketed by arm() / disarm(); a deadline that passes means a peer is dead or
    hung -- rank 0 prints the error line (once; never after the real line), every rank leaves with os._exit so that nothing
    waits in a destructor.  SIGTERM (torch.distributed.run tearing the job down after a peer's failure) does the same."""

    def __init__(self, rank: int, world: int):
        import threading
        self.rank, self.world, self.done, self.timer, self._lock = rank, world, False, None, threading.Lock()
        self.default_s = float(os.environ.get("CRESTE_BENCH_BARRIER_TIMEOUT_S", "600"))
        if world > 1 or os.environ.get("CRESTE_BENCH_LAUNCHED") == "1":
            import signal
            signal.signal(signal.SIGTERM, lambda *_: self.fail("terminated by the launcher (a peer rank failed or the job timed out)", 143))

    def arm(self, what: str, seconds: float | None = None):
        import threading
        self.disarm()
        secs = self.default_s if seconds is None else seconds
        self.timer = threading.Timer(secs, self.fail, (f"rank {self.rank}: '{what}' did not complete within {secs:.0f} s "
                                                       f"(a peer rank is dead or hung)", 124))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def fail(self, message: str, code: int):
        with self._lock:
            if self.done:
                os._exit(code)
            self.done = True
        if self.rank == 0:
            print(json.dumps(error_line(self.world, message)), flush=True)
        else:
            print(f"[bench rank {self.rank}] {message}", file=sys.stderr, flush=True)
        os._exit(code)

    def printed(self):
        with self._lock:
            self.done = True
        self.disarm()


def dry_run(args) -> int:
    """The launcher + timing bookkeeping of `main` with nothing to run on: process group on gloo, a step = a sleep of 2 ms x
    (rank + 1), barrier + max over ranks around exactly K steps, ONE line from rank 0.  For the CPU test of `--gpus N`."""
    import torch.distributed as dist
    from creste_public_amd import dist_utils
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    dog = RankWatchdog(rank, world)
    die = os.environ.get("CRESTE_BENCH_TEST_DIE", "")            # (tests: a rank that dies / hangs before its first barrier)
    if dist.is_initialized() and die == str(rank):
        os._exit(17)
    if dist.is_initialized() and die == f"sleep:{rank}":
        time.sleep(3600)
    for _ in range(args.warmup):
        time.sleep(0.002)
    if dist.is_initialized():
        dog.arm("barrier before the timed steps")
        dist.barrier()
        dog.disarm()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if dist.is_initialized():
        dog.arm("barrier behind the timed steps")
        dist.barrier()
        dog.disarm()
    el = dist_utils.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"metric": "frames/sec (RGB+LiDAR->BEV costmap)", "value": round(args.batch * world * args.steps / el, 3),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "data": "dry-run", "launched_by": "self" if os.environ.get("CRESTE_BENCH_LAUNCHED") else "external",
                          "note": "NOT a measurement: launcher check, a step is a sleep"}), flush=True)
    dog.printed()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16x6", choices=["f32", "bf16x6", "f16x3", "bf16x3", "bf16"],
                    help="operand precision of the convs on the matrix cores; the headline runs the fp32-equivalent "
                         "bf16x6 (the reference computes in fp32 end to end)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (H2D inside the timed region) run")
    ap.add_argument("--no-irl", action="store_true", help="skip the IRL train-step timing")
    ap.add_argument("--no-modes", action="store_true", help="skip the short extra runs of the other precisions")
    ap.add_argument("--parts", type=int, default=-1,
                    help="pipelined inference: the batch as this many forwards on as many streams (MaxEntIRL.inference_parts); "
                         "-1 = the model's default (2), 0 / 1 = one forward on one stream")
    ap.add_argument("--layers", default="", help="write a per-conv-shape timing table to this file")
    ap.add_argument("--no-train-dp", action="store_true",
                    help="under torch.distributed.run: skip the data-parallel training legs (configs[3]/[4])")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / bookkeeping check without a GPU (tests/test_dist_cpu.py): gloo, a sleep for a step; "
                         "the line says data = 'dry-run' and is NOT a measurement")
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("CRESTE_BENCH_FORCE_LAUNCH") == "1") and "RANK" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: become N ranks, one per GPU, under torch.distributed.run on this
        # node; rank 0 of the children prints the single JSON line, this process only passes output and exit code through
        raise SystemExit(self_launch(args.gpus))
    if args.dry_run:
        raise SystemExit(dry_run(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):   # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        from creste_public_amd import dist_utils as _du
        _du.init_rccl(torch.device("cuda", local_rank))       # (collectives on a high-priority stream: see its docstring)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dog = RankWatchdog(rank, world)
    if os.environ.get("CRESTE_BENCH_TEST_DIE") == str(rank) and dist is not None:      # (tests: a rank that dies before its first barrier)
        os._exit(17)

    import creste_public_amd
    from creste_public_amd import synth
    creste_public_amd.set_precision(args.precision)
    model = build_model(device)
    if args.parts >= 0:
        model.inference_parts = args.parts
    from creste_public_amd.creste.utils.projection import lidar_depth_images
    gen = torch.Generator().manual_seed(1337 + rank)
    rgbd = torch.zeros(args.batch, 1, 4, IMG_H, IMG_W, device=device)
    rgbd[:, 0, :3] = torch.rand(args.batch, 3, IMG_H, IMG_W, generator=gen).to(device)   # RGB in [0,1)
    scan = synth.lidar_scan(args.batch, gen).to(device)               # [B, 128*1024, 3] LiDAR points
    l2c = synth.lidar2camrect(args.batch, IMG_H, IMG_W).to(device)    # float64 projection
    p2p = synth.make_p2p(args.batch, IMG_H, IMG_W).to(device)

    def step(rgbd=rgbd, scan=scan):
        """RGB + LiDAR scan -> costmap: project the scan into the sparse millimetre depth channel of the
        RGB-D tensor (HIP scatter-max), then the perception -> BEV -> costmap forward."""
        with torch.no_grad():
            lidar_depth_images(scan, l2c, IMG_H, IMG_W, out=rgbd[:, 0, 3], scale=1000.0, depth_priority="max")
            return model((rgbd, p2p))

    for _ in range(args.warmup):
        out = step()
    with torch.no_grad():
        parts_used = model._parts_for(args.batch)
    # The timed region of a pipelined run carries NO instrumentation: per-kernel event pairs there (two per conv launch and
    # part, ~400 timing events per step on two streams) are not a property of the kernels (they span the other part's
    # kernels) and were seen to disturb the very steps they sit in (single runs of 42-57 ms per step on boxes whose host-fed
    # loop -- same steps, no events -- read 38-39).  Per-kernel numbers come from the one-stream steps right behind it.
    prof = ConvProfiler()
    if parts_used == 1:
        prof.install()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dog.arm("barrier around the timed steps")
            dist.barrier()
            dog.disarm()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if parts_used == 1:
        prof.uninstall()
    assert torch.isfinite(out["traversability_preds"]).all()
    # Per-KERNEL numbers (roofline, roofline_splat, the GEMM probe, --layers) come from the same number of steps run as ONE
    # forward on ONE stream right after the timed region: inside the pipelined steps the parts' kernels share the device, an
    # event pair around a launch also spans the other part's kernels, and a per-launch duration is not a property of that
    # kernel any more
    serial_ms = None
    if parts_used > 1:
        keep = model.inference_parts
        model.inference_parts = 0
        step()
        prof = ConvProfiler()
        prof.install()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out1 = step()
        fence()
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        prof.uninstall()
        gemm_probe = gemm_kernel_probe(step) if rank == 0 else None
        model.inference_parts = keep
        step()
        del out1
    else:
        gemm_probe = gemm_kernel_probe(step) if rank == 0 else None
    from creste_public_amd import dist_utils
    elapsed = dist_utils.max_over_ranks(elapsed, device)     # the job is as slow as its slowest rank

    host_fed = None
    if not args.no_host_fed:
        # the same steps with every batch arriving from PINNED HOST memory inside the timed region: RGB-D frames
        # (channel 3 = the depth plane the LiDAR kernel fills) + the LiDAR scan, copied on a copy stream into one of two
        # device buffers while the previous batch computes
        h_rgbd, h_scan = rgbd.cpu().pin_memory(), scan.cpu().pin_memory()
        bufs = [(torch.empty_like(rgbd), torch.empty_like(scan)) for _ in range(2)]
        cstream, cur = torch.cuda.Stream(device=device), torch.cuda.current_stream()
        ready = [torch.cuda.Event() for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]

        def enqueue_copy(k):
            with torch.cuda.stream(cstream):
                cstream.wait_event(free[k])              # the step that last read buffer k has finished
                bufs[k][0].copy_(h_rgbd, non_blocking=True)
                bufs[k][1].copy_(h_scan, non_blocking=True)
                ready[k].record(cstream)

        def host_fed_steps(n):
            enqueue_copy(0)
            for i in range(n):
                k = i & 1
                if i + 1 < n:
                    enqueue_copy(k ^ 1)
                cur.wait_event(ready[k])
                o = step(*bufs[k])
                free[k].record(cur)
            return o
        for k in range(2):
            free[k].record(cur)
        host_fed_steps(2)
        fence()
        t1 = time.perf_counter()
        out_h = host_fed_steps(args.steps)
        fence()
        el_h = dist_utils.max_over_ranks(time.perf_counter() - t1, device)
        # the host-fed steps must reproduce the resident ones bit for bit: a mismatch means frames were corrupted somewhere in
        # the timed steps, and a throughput of wrong frames is not a measurement -- the run FAILS, after saying
        # which frames differ and whether a fresh step sides with the resident or the host-fed result
        same_as_resident = torch.equal(out_h["traversability_preds"], out["traversability_preds"])
        if not same_as_resident:
            again = step()
            rows = lambda a, b: torch.nonzero((a != b).reshape(a.shape[0], -1).any(1)).flatten().tolist()   # noqa: E731
            mismatch = {k: {"host_fed_vs_timed": rows(out_h[k], out[k]), "fresh_vs_timed": rows(again[k], out[k]),
                            "fresh_vs_host_fed": rows(again[k], out_h[k])} for k in out if out[k].is_floating_point()}
            print(f"[bench] FAILED: the host-fed step differs from the resident one: {mismatch}", file=sys.stderr, flush=True)
            raise SystemExit(3)
        host_fed = {"value": round(args.batch * args.gpus * args.steps / el_h, 3), "ms_per_step": round(el_h / args.steps * 1e3, 3),
                    "h2d_bytes_per_step": int(h_rgbd.numel() * 4 + h_scan.numel() * 4),
                    "note": "pinned host batch (RGB-D frames + LiDAR scan) -> device on a copy stream, two device buffers: "
                            "the copy of batch k+1 overlaps the compute of batch k; the first copy of the timed region "
                            "is exposed; `equals_resident`: the last host-fed step's costmap == the resident run's, bit for bit",
                    "equals_resident": bool(same_as_resident)}
        del bufs, h_rgbd, h_scan

    modes = {}
    if args.gpus == 1 and not args.no_modes:
        # every other conv operand mode over the SAME number of steps after 1 warm-up (same model, same inputs)
        for name in ("f32", "bf16x6", "f16x3", "bf16x3", "bf16"):
            if name == args.precision:
                continue
            creste_public_amd.set_precision(name)
            step(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            modes[name] = round(args.steps * args.batch / (time.perf_counter() - t1), 2)
        creste_public_amd.set_precision(args.precision)

    if rank == 0 and args.layers:
        per = {}
        for e0, e1, fl, bn, shape in prof.records:
            d = per.setdefault((shape, bn), [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += 1
        with open(args.layers, "w") as f:
            f.write("Cin,Cout,K,Ho,Wo,kernel,calls_per_step,ms_per_step,TFLOPs\n")
            for (shape, bn), (ms, fl, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
                f.write(",".join(map(str, shape)) + f",{bn[1].replace(',', ';')},{n / args.steps:.1f},{ms / args.steps:.4f},"
                        f"{fl / (ms * 1e-3) / 1e12:.2f}\n")
    if rank == 0:
        frames = args.batch * args.gpus * args.steps
        by = prof.summary()
        dom = max(by, key=lambda k: by[k]["ms"])
        d = by[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        dprec, kname = dom
        conv_ms = sum(v["ms"] for v in by.values())
        traffic, traffic_note, mfma_busy = None, None, None
        pj = os.path.join(ROOT, "profiles", "roofline_counters.json")
        if os.path.exists(pj):
            try:
                # HBM bytes of one F(4x4,3x3) conv call = the PMC bytes of EVERY kernel of the family (both input-transform
                # instantiations, the GEMM, the output transform and the fused output -> input transform of the conv pairs)
                # over the profiled run / the family's GEMM launches (every call launches exactly one GEMM)
                cnt = json.load(open(pj))
                fam = {k: v for k, v in cnt.items() if isinstance(v, dict) and k.startswith("wino4_") and "wg_" not in k and "pack" not in k}
                gemms = sum(v["launches"] for k, v in fam.items() if "gemm32" in k)
                if gemms:
                    traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam.values()) / gemms
                    traffic_note = (f"CONSTANT of the committed profile, not of this run: PMC (profiles/roofline_counters.json: {cnt.get('_source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --parts 1')}; "
                                    f"commit {cnt.get('_commit')}, box {cnt.get('_box')}; {cnt.get('_launch_check', 'launch counts unchecked')}): "
                                    f"HBM bytes of all {len(fam)} kernels of the F(4x4,3x3) family / {gemms} conv calls, i.e. the average over "
                                    "the step's calls of both GEMM tile widths")
            except Exception:
                traffic = None
        import glob
        pms = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_encoder.json")))      # the newest round's table
        pm = pms[-1] if pms else ""
        if pm and os.path.exists(pm):
            try:
                mfma_busy = json.load(open(pm))
            except Exception:
                mfma_busy = None
        line = {
            "metric": "frames/sec (RGB+LiDAR->BEV costmap)",
            "value": round(frames / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic",
            "config": {"workload": f"inference: batch={args.batch}/GPU synthetic {IMG_W}x{IMG_H} RGB + 128x1024 "
                                   "LiDAR scan (projected to the sparse depth channel inside the step) -> 256x256 BEV costmap "
                                   "(MaxEntIRL solve_mdp=False; EfficientNet-B0 U-Net, BEV splat, ResNet-18 heads, "
                                   "reward FCN), random-init weights",
                       "batch_per_gpu": args.batch, "image": [IMG_H, IMG_W], "lidar": [128, 1024],
                       "bev": [256, 256], "parallelism": f"{args.gpus} independent replicas (frame-sharded, no collective)",
                       "pipeline": (f"every step = the batch's {parts_used} parts of {args.batch // parts_used} frames, each the "
                                    f"whole forward on its own HIP stream, one host thread (MaxEntIRL.inference_parts = "
                                    f"{parts_used}; outputs in shared whole-batch buffers)") if parts_used > 1
                                   else "one forward on one stream",
                       "inputs": "resident in HBM when the timed region starts (`value`); `value_host_fed` = the same steps "
                                 "with every 214 MB batch copied from pinned host memory inside the timed region"},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": round(PEAK[dprec], 1), "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[dprec], 4), "traffic": traffic, "traffic_note": traffic_note,
                         "mfma_busy_gemm": (mfma_busy or {}).get("gemm"), "mfma_busy_encoder": (mfma_busy or {}).get("encoder"),
                         "mfma_busy_whole_step": (mfma_busy or {}).get("whole_step"),
                         "mfma_busy_note": (mfma_busy or {}).get("note"),
                         "mfma_products_per_multiply": round(PRODUCTS[dprec], 3),
                         "mfma_issue_util": round(PRODUCTS[dprec] * achieved / PEAK[dprec], 4),
                         "launches": d["n"], "avg_launch_ms": round(d["ms"] / d["n"], 4),
                         "conv_share_of_step": round(conv_ms / args.steps / (serial_ms or elapsed / args.steps * 1e3), 4),
                         "note": "achieved = algorithmic conv FLOPs (2*M*Cout*Cin*K*K, the DIRECT conv's count) of every "
                                 "launch of this kernel symbol in the timed steps / their HIP-event time (a Winograd call "
                                 "= its GEMM + output-transform kernels together); peak = dense peak of the MFMA "
                                 "instruction issued; mfma_issue_util counts the piece products actually issued"},
        }
        if gemm_probe is not None:
            line["roofline"]["gemm_kernel"] = gemm_probe
        sr = prof.splat_roofline(keyed_geometry_extra_ms(device))
        if sr is not None:
            try:      # PMC bytes of the step's splat kernels (profiles/roofline_counters.json, calibrated WRITE_SIZE / FETCH_SIZE)
                cnt = json.load(open(os.path.join(ROOT, "profiles", "roofline_counters.json")))
                ks = [k for k in cnt if isinstance(cnt[k], dict) and k.startswith("splat_") and "launches" in cnt[k]]
                if any("splat_gather8" in k for k in ks):
                    sr["traffic"] = round(sum(cnt[k]["hbm_bytes_per_launch"] for k in ks), 1)
                    sr["traffic_kernels"] = {k: round(cnt[k]["hbm_bytes_per_launch"] / 1e6, 1) for k in ks}
                    sr["traffic_note"] = ("PMC MB per launch of every splat kernel of a step; " + str(cnt.get("_note", "")) +
                                          " -- the gather's 402.7 MB of BEV rows leave as nontemporal stores, which WRITE_SIZE "
                                          "under-reports (calibrated against an empty plan: profiles/r05_pmc_calibration.json)")
            except Exception:
                pass
            line["roofline_splat"] = sr
        if serial_ms is not None:
            line["ms_per_step_one_stream"] = round(serial_ms, 3)
            line["roofline"]["measured_over"] = (f"{args.steps} steps run as one forward on one stream ({serial_ms:.2f} ms / step) "
                                                 "right after the timed region -- stand-alone launch durations, the ones "
                                                 "`rocprofv3 --kernel-trace` of `bench.py --parts 1` shows; the timed "
                                                 "(pipelined) steps themselves carry no event pairs")
        if host_fed is not None:
            line["value_host_fed"] = host_fed["value"]
            line["host_fed"] = host_fed
        if modes:
            line["modes_frames_per_s"] = dict(modes, **{args.precision: line["value"]})
            line["modes_note"] = (f"every mode timed over the same {args.steps} steps after 1 warm-up; " +
                                  "; ".join(f"{k} = {v}" for k, v in DTYPE.items()))
            line["fp32_equivalent_frames_per_s"] = {k: line["modes_frames_per_s"][k] for k in ("f32", "bf16x6")}
            line["narrower_than_fp32_frames_per_s"] = {k: line["modes_frames_per_s"][k] for k in ("f16x3", "bf16x3", "bf16")}
        if args.gpus == 1 and not args.no_irl:
            line["latency"] = latency_extras(model, device)
            line["irl"] = irl_extras(model, device)
            # the same steps with the narrower f16x3 operands (fp16 hi+lo = 22 significand bits): side entries
            if args.precision != "f16x3":
                creste_public_amd.set_precision("f16x3")
                try:
                    m16 = build_model(device)
                    line["irl_f16x3"] = {k: irl_step_bench(m16, device, k) for k in ("reference", "mdp256")}
                    line["distill_f16x3"] = distill_extras(device)
                    line["ssc_f16x3"] = ssc_extras(device)
                    del m16
                finally:
                    creste_public_amd.set_precision(args.precision)
            vi = line["irl"]["vi_8x256x256"]
            line["roofline_vi"] = {"bound": "hbm", "kernel": "creste_value_iteration_f32 (all sweeps + q / policy read-out)",
                                   "achieved": vi["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": vi["frac_of_hbm_peak"], "ms": vi["ms"], "sweeps": vi["sweeps"], "traffic": None,
                                   "note": "r ~ U[0,1) [8,256,256], gamma 0.99, threshold 1e-3; bytes = B*H*W*(12*sweeps + 72) "
                                           "(SURVEY 8d: the HBM-streaming figure of a sweep-per-launch solver; the state is "
                                           "LDS / L2 resident here, so the fraction may exceed 1)"}
            # The solver keeps its state in registers / LDS: what bounds a sweep is VALU issue (117 VALU instructions per
            # 1 x 4 strip and sweep, counted in the ISA of vi_spec_kernel; a wave64 instruction holds a SIMD for 4 cycles)
            # and LDS traffic (two 16-byte reads + one 16-byte write per strip and sweep), not HBM
            def vi_onchip(ms, sweeps, nwg, waves, strips):
                clk, simds, cus = 2.4e9, 1024, 256
                t = ms * 1e-3
                valu = nwg * waves * 117 * 4.0 * sweeps / (simds * clk * t)
                lds_bytes = nwg * strips * 48.0 * sweeps
                return {"valu_issue_frac": round(valu, 4), "lds_GBps": round(lds_bytes / t / 1e9, 1),
                        "lds_frac": round(lds_bytes / t / (cus * 128 * clk), 4)}
            line["roofline_vi"]["on_chip"] = dict(vi_onchip(vi["ms"], vi["sweeps"], 256, 16, 48 * 20),
                note="8x256x256: 256 workgroups (32 x 64 tiles + 8-cell halo = 48 x 80 cells), 16 waves each; valu_issue_frac = "
                     "waves x 117 VALU instructions x 4 cycles x sweeps / (1024 SIMDs x 2.4 GHz x time): the bound a sweep is "
                     "closest to (the rest: barrier / LDS / DPP latency inside a sweep and the halo exchange, profiles/r04_value_iteration.md); "
                     "lds_frac against 256 CUs x 128 B/clk")
            vr = line["irl"]["vi_8x64x128"]
            line["roofline_vi"]["on_chip_8x64x128"] = vi_onchip(vr["ms"], vr["sweeps"], 256, 4, 32 * 8)
            line["roofline_vi"]["reference_grid_8x64x128"] = {"achieved": vr["algorithmic_GBps"], "frac": vr["frac_of_hbm_peak"],
                                                              "ms": vr["ms"], "sweeps": vr["sweeps"]}
            line["distill"] = distill_extras(device)
            line["ssc"] = ssc_extras(device)
    train_dp = None
    if dist is not None and not args.no_train_dp:
        # every rank takes part (synchronous SGD); rank 0 reports
        dog.arm("data-parallel training legs", 900)
        train_dp = train_dp_extras(model, device, rank)
        dog.disarm()
    if dist is not None:
        torch.cuda.synchronize()
        dog.arm("host-side gloo group")
        host_group = dist.new_group(backend="gloo") if world > 1 else None
        dog.disarm()
    if rank == 0:
        if train_dp is not None:
            line["train_dp"] = train_dp
        if not args.no_cpu_baseline:
            # rank 0 only, at every N, after all GPU work: the other ranks wait on the HOST (a gloo barrier: no collective
            # kernel spins on their GPUs meanwhile); `cores` says how many host threads the baseline used
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
        dog.printed()
    if dist is not None:
        if rank != 0:
            dog.arm("waiting for rank 0's CPU baseline", 1200)       # (~250 s of oracle forwards on rank 0's host cores)
        if host_group is not None:
            dist.barrier(group=host_group)
        dist.barrier()
        dog.printed()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
