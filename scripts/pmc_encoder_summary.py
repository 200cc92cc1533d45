"""Summary of scripts/pmc_encoder.sh: per kernel family of ONE inference step (the last lidar_depth_kernel .. end of the first
pass of bench.py --parts 1), SQ_VALU_MFMA_BUSY_CYCLES against the SIMD-cycles the kernel had: GRBM_GUI_ACTIVE / 8 XCDs x 1024
SIMDs.  Encoder = every kernel from nchw4_to_nhwc (the RGB-D frame enters) up to pixel_geometry (its features leave)."""
import csv, glob, re, sys, collections
out = sys.argv[1]


def load(sub):
    rows = []
    for f in glob.glob(f"{out}/{sub}/**/pmc_counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        d = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


sq, gr = load("sq"), load("grbm")


def one_step(rows):
    """the dispatches of the 2nd step() of the run: between the 2nd and 3rd lidar_depth_kernel"""
    marks = [i for i, r in enumerate(rows) if "lidar_depth_kernel" in r["name"]]
    return rows[marks[1]:marks[2]]


a, b = one_step(sq), one_step(gr)
assert [r["name"] for r in a] == [r["name"] for r in b], "the two passes dispatched different kernel sequences"


def fam(n):
    n = re.sub(r"^void ", "", n).replace("creste::", "")
    return re.sub(r"\(.*", "", n)[:60]


enc_lo = next(i for i, r in enumerate(a) if "nchw4_to_nhwc" in r["name"])
enc_hi = next(i for i, r in enumerate(a) if "pixel_geometry" in r["name"])
tot = collections.OrderedDict()
for i, (s, g) in enumerate(zip(a, b)):
    for scope in (fam(s["name"]), "ENCODER (nchw4_to_nhwc .. pixel_geometry)" if enc_lo <= i < enc_hi else None, "WHOLE STEP"):
        if scope is None:
            continue
        d = tot.setdefault(scope, collections.Counter())
        d["n"] += 1
        d["mfma_busy"] += s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        d["mfma_insts"] += s.get("SQ_INSTS_MFMA", 0.0)
        d["simd_cycles"] += g.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
        d["gui"] += g.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
print("# MFMA-busy per kernel family of one inference step (bench.py --parts 1, batch 16, bf16x6), rocprofv3 --pmc")
print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); kernels run serialised under --pmc")
print(f"# {'kernel family':62s} {'calls':>5s} {'Mcycles':>9s} {'MFMA busy':>10s} {'MFMA insts':>12s}")
rows = sorted(tot.items(), key=lambda kv: -kv[1]["gui"])
for k, d in rows:
    if d["simd_cycles"] <= 0:
        continue
    print(f"  {k:62s} {int(d['n']):5d} {d['gui'] / 1e6:9.2f} {d['mfma_busy'] / d['simd_cycles']:10.3f} {d['mfma_insts']:12.4g}")

import json
def _frac(name):
    d = next((v for k, v in tot.items() if k.startswith(name)), None)
    return round(d["mfma_busy"] / d["simd_cycles"], 4) if d and d["simd_cycles"] > 0 else None
json.dump({"gemm": _frac("wino4_gemm32_kernel<3, 4"), "gemm_128_cout_tile": _frac("wino4_gemm32_kernel<3, 2"),
           "encoder": _frac("ENCODER"), "whole_step": _frac("WHOLE STEP"),
           "note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per kernel family of ONE inference step of "
                   "bench.py --parts 1 (batch 16, bf16x6), two rocprofv3 --pmc passes (scripts/pmc_encoder.sh, table: "
                   "profiles/<tag>_pmc_encoder.txt); gemm = wino4_gemm32_kernel<3, 4> (256-cout tiles, 18 of the 19 calls), encoder = "
                   "every kernel from nchw4_to_nhwc to pixel_geometry; a CONSTANT of the committed profile, not of the run that prints it"},
          open(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r06_pmc_encoder.json", "w"), indent=1)
