"""Which kernel misbehaves when two half-batches run concurrently on two streams?  Every stage of the encoder is run on
its serially computed input, on two streams at once (different halves), R times without synchronisation, and compared with
its serial output."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
import creste_public_amd
from creste_public_amd import synth, ops
B, H, W = 16, bench.IMG_H, bench.IMG_W
device = torch.device("cuda", 0)
creste_public_amd.set_precision("bf16x6")
model = bench.build_model(device)
model.inference_parts = 0
rgbd, p2p = synth.make_frames(B, H, W, seed=99)
rgbd = rgbd.to(device)
h = B // 2
eff = model.backbone.depthcomp.depthcomp.vision_backbone
eff = getattr(eff, "model", eff)
print(type(eff).__name__)
trunk = eff.trunk
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def clone(o):
    if isinstance(o, ops.Act):
        return o.buf[..., o.co:o.co + o.C].clone()
    if isinstance(o, (tuple, list)):
        return [clone(v) for v in o]
    if isinstance(o, dict):
        return {k: clone(v) for k, v in o.items()}
    return o.clone() if torch.is_tensor(o) else o


def same(a, b):
    if isinstance(a, (list, tuple)):
        return all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a)
    return torch.equal(a, b) if torch.is_tensor(a) else True


def check(name, fn, ins):
    """fn(input of half i) -> output; ins = [input of half 0, input of half 1]"""
    with torch.no_grad():
        refs = [clone(fn(x)) for x in ins]
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        outs = [[], []]
        for s in streams:
            s.wait_stream(main)
        for r in range(R):
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    outs[i].append(fn(ins[i]))
        torch.cuda.synchronize()
        bad = sum(not same(clone(o), refs[i]) for i in range(2) for o in outs[i])
        print(f"{name:50s} {'ok' if not bad else f'DIFFERS in {bad} of {2 * R} runs'}", flush=True)
        return [fn(x) for x in ins]


x = [ops.nchw_to_nhwc(rgbd[i * h:(i + 1) * h].reshape(h, 4, H, W).contiguous()) for i in range(2)]
check("whole encoder (trunk + Up blocks + 1x1)", lambda a: eff.forward_act(a)[0], x)
eps = check("EfficientNet trunk", lambda a: trunk.extract_endpoints_act(a), x)
# block by block on serially computed inputs
with torch.no_grad():
    b0, pad = trunk._blocks[0], trunk._conv_stem.static_pad
    H1, W1 = (H + pad[0] + pad[1] - 3) // 2 + 1, (W + pad[2] + pad[3] - 3) // 2 + 1
    ws, bs = trunk._stem_fused.get()
    wd, bd = b0._plan["dw"].get()
    se = b0._plan["se"].get()


def stem(a):
    hh, gate = ops.stem_dw_se(a, ws, bs, pad, wd, bd, b0._depthwise_conv.static_pad, *se)
    return hh, gate


hg = check("stem + depthwise + SE sums / gate", stem, x)
cur = check("block 0 project (gated 1x1)", lambda t: b0.project_act(None, t[0], t[1]), hg)
for i, blk in enumerate(trunk._blocks):
    if i == 0:
        continue
    cur = check(f"MBConv block {i} (k{blk.k} s{blk.s} {blk.cin}->{blk.cout})", blk.forward_act, cur)
hcur = [e["reduction_5"] for e in eps]
for i in range(1, eff.n_ups + 1):
    up = getattr(eff, f"up{i}")
    pairs = [(hcur[j], eps[j][f"reduction_{5 - i}"]) for j in range(2)]
    hcur = check(f"Up block {i}", lambda t: up.forward_act(t[0], t[1]), pairs)
check("final 1x1", lambda a: eff._final(a), hcur)

# inside the two stride-2 fused blocks: the fused expand + depthwise + SE kernel pair, then the gated project conv
from creste_public_amd.creste.models.blocks import effnet as E
with torch.no_grad():
    x = [ops.nchw_to_nhwc(rgbd[i * h:(i + 1) * h].reshape(h, 4, H, W).contiguous()) for i in range(2)]
    hg = [stem(a) for a in x]
    cur = [b0.project_act(None, t[0], t[1]) for t in hg]
for i, blk in enumerate(trunk._blocks):
    if i == 0:
        continue
    if i in (1, 2, 3):
        p = blk._plan
        w, b = p["dw"].get()
        pad = blk._depthwise_conv.static_pad
        we, be = p["expand_fused"].get()
        sew = p["se"].get()
        fused = lambda a, blk=blk: ops.mbconv_expand_dw_se(a, we, be, w, b, blk.k, blk.s, pad, *sew)
        hg = check(f"  block {i}: fused expand+dw+SE (h, gate)", fused, cur)
        only_h = lambda a, blk=blk: ops.mbconv_expand_dw_se(a, we, be, w, b, blk.k, blk.s, pad, *sew)[0]
        check(f"  block {i}: fused expand+dw (h only)", only_h, cur)
        pairs = [(cur[j], hg[j]) for j in range(2)]
        check(f"  block {i}: gated project", lambda t, blk=blk: blk.project_act(t[0], t[1][0], t[1][1]), pairs)
    with torch.no_grad():
        cur = [blk.forward_act(c) for c in cur]


def pair(name, fa, xa, fb, xb, reps=12):
    with torch.no_grad():
        ra, rb = clone(fa(xa)), clone(fb(xb))
        torch.cuda.synchronize()
        oa, ob = [], []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for r in range(reps):
            with torch.cuda.stream(streams[0]):
                oa.append(fa(xa))
            with torch.cuda.stream(streams[1]):
                ob.append(fb(xb))
        torch.cuda.synchronize()
        ba = sum(not same(clone(o), ra) for o in oa)
        bb = sum(not same(clone(o), rb) for o in ob)
        print(f"{name:60s} A wrong {ba}/{reps}   B wrong {bb}/{reps}", flush=True)


print("pairs (stream 0 runs A on half 0, stream 1 runs B on half 1):")
with torch.no_grad():
    x = [ops.nchw_to_nhwc(rgbd[i * h:(i + 1) * h].reshape(h, 4, H, W).contiguous()) for i in range(2)]
    hg0 = [stem(a) for a in x]
    cur = [b0.project_act(None, t[0], t[1]) for t in hg0]
for i in (1, 3):
    blk = trunk._blocks[i]
    if i == 3:
        with torch.no_grad():
            cur = [trunk._blocks[2].forward_act(trunk._blocks[1].forward_act(c)) for c in cur]
    p = blk._plan
    w, b = p["dw"].get()
    pad = blk._depthwise_conv.static_pad
    we, be = p["expand_fused"].get()
    sew = p["se"].get()
    fused = lambda a, blk=blk, we=we, be=be, w=w, b=b, pad=pad, sew=sew: ops.mbconv_expand_dw_se(a, we, be, w, b, blk.k, blk.s, pad, *sew)
    with torch.no_grad():
        hg = [fused(c) for c in cur]
    proj = lambda t, blk=blk: blk.project_act(None, t[0], t[1])
    pair(f"block {i}: A = fused, B = fused", fused, cur[0], fused, cur[1])
    pair(f"block {i}: A = project, B = project", proj, hg[0], proj, hg[1])
    pair(f"block {i}: A = fused, B = project", fused, cur[0], proj, hg[1])
    pair(f"block {i}: A = whole block, B = whole block", blk.forward_act, cur[0], blk.forward_act, cur[1])
    keep = []
    def whole_keep(a, blk=blk, fused=fused):
        t = fused(a); keep.append(t)
        return blk.project_act(None, t[0], t[1])
    pair(f"block {i}: whole block, intermediates kept alive", whole_keep, cur[0], whole_keep, cur[1])
    keep.clear()

print("where does the fused kernel's output differ when the gated project conv runs beside it?")
blk = trunk._blocks[1]
with torch.no_grad():
    x = [ops.nchw_to_nhwc(rgbd[i * h:(i + 1) * h].reshape(h, 4, H, W).contiguous()) for i in range(2)]
    hg0 = [stem(a) for a in x]
    cur = [b0.project_act(None, t[0], t[1]) for t in hg0]
    p = blk._plan
    w, b = p["dw"].get(); pad = blk._depthwise_conv.static_pad
    we, be = p["expand_fused"].get(); sew = p["se"].get()
    fused = lambda a: ops.mbconv_expand_dw_se(a, we, be, w, b, blk.k, blk.s, pad, *sew)
    proj = lambda t: blk.project_act(None, t[0], t[1])
    hgB = fused(cur[1])
    ref = fused(cur[0]); rh, rg = ref[0].buf.clone(), ref[1].clone()
    torch.cuda.synchronize()
    oa = []
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for r in range(12):
        with torch.cuda.stream(streams[0]):
            oa.append(fused(cur[0]))
        with torch.cuda.stream(streams[1]):
            proj(hgB)
    torch.cuda.synchronize()
    for r, (hh, gg) in enumerate(oa):
        dh = (hh.buf != rh) | (hh.buf.isnan() != rh.isnan())
        dg = gg != rg
        if dh.any() or dg.any():
            idx = torch.nonzero(dh)
            msg = f"run {r}: h differs in {int(dh.sum())} values, gate in {int(dg.sum())}"
            if len(idx):
                lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
                v = hh.buf[dh][:6].tolist(); rv = rh[dh][:6].tolist()
                msg += f"; h index range n,y,x,c {lo}..{hi}; got {v} want {rv}; nan {int(hh.buf.isnan().sum())}"
                ys = sorted(set(idx[:, 1].tolist()))[:20]; xs = sorted(set(idx[:, 2].tolist()))[:40]
                msg += f"; rows {ys}; cols {xs}"
            print(msg)

print("which neighbours disturb block 1's fused kernel?")
with torch.no_grad():
    big = ops.Act(torch.randn(8, 152, 304, 256, device="cuda"), 256)
    wgt = torch.randn(256, 256, 3, 3, device="cuda") / 48
    pcw = ops.pack_conv(wgt, None, None, 1, 1, ops.ACT_RELU, ops.PREC_BF16X6, algo=ops.ALGO_WINOGRAD4)
    w1 = torch.randn(64, 256, 1, 1, device="cuda") / 16
    pc1 = ops.pack_conv(w1, None, None, 1, 0, ops.ACT_NONE, ops.PREC_BF16X6)
    img = torch.randn(8, 4, 608, 1216, device="cuda")
    t1 = torch.randn(64 << 20, device="cuda")
pair("block 1: A = fused, B = F(4x4) conv 256->256", fused, cur[0], lambda a: ops.conv2d(a, pcw), big)
pair("block 1: A = fused, B = plain 1x1 conv 256->64 (no gate)", fused, cur[0], lambda a: ops.conv2d(a, pc1), big)
pair("block 1: A = fused, B = nchw->nhwc copy", fused, cur[0], lambda a: ops.nchw_to_nhwc(a), img)
pair("block 1: A = fused, B = torch elementwise", fused, cur[0], lambda a: a * 2.0 + 1.0, t1)
