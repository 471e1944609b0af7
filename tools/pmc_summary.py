#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc runs of bench.py per kernel (development aid; writes the JSON kept under profiles/).

    python tools/pmc_summary.py traffic <fetch_dir> <write_dir> <out.json> <generate_calls>
    python tools/pmc_summary.py sq <dir> <out.json>

traffic: FETCH_SIZE and WRITE_SIZE come from separate passes (they do not fit one); counter unit KB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads, so hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def counters(d):
    """{kernel: {counter: [per-dispatch values]}} from the counter_collection csv under d."""
    out = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    for f in files:
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def conv_launch(name):
    return ("igemm_kernelIDF16b" in name and ("Li128ELi" in name or "Li256ELi" in name)) or "igemm8_kernel" in name or "fused_block_kernel" in name or "stem_pool_kernel" in name


def traffic(fetch_dir, write_dir, out, calls):
    f, w = counters(fetch_dir), counters(write_dir)
    res = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 1 --warmup 1 "
                   f"--cpu-images 0 --concurrent-steps 0` ({calls} generate calls). Counter unit KB. On gfx950 FETCH_SIZE reports half "
                   "the bytes of wide coalesced reads; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.", "kernels": {}}
    agg = [0, 0.0, 0.0]
    for k in sorted(set(f) | set(w)):
        fv, wv = f[k].get("FETCH_SIZE", []), w[k].get("WRITE_SIZE", [])
        n = max(len(fv), len(wv))
        if n == 0:
            continue
        fk, wk = sum(fv) / max(1, len(fv)), sum(wv) / max(1, len(wv))
        res["kernels"][k] = {"launches": n, "fetch_kb_per_launch": fk, "write_kb_per_launch": wk,
                             "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
        if conv_launch(k):
            agg[0] += n; agg[1] += sum(fv); agg[2] += sum(wv)
    res["igemm_bf16_128row_tiles"] = {
        "what": "igemm_kernel<bf16> launches with tiles of 128 rows and more plus fused_block_kernel and stem_pool_kernel launches (the encoder's convolutions)",
        "launches": agg[0], "hbm_bytes_per_launch": (2 * agg[1] + agg[2]) * 1024 / max(1, agg[0]),
        "fetch_gb_per_generate_corrected": 2 * agg[1] * 1024 / calls / 1e9, "write_gb_per_generate": agg[2] * 1024 / calls / 1e9}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["igemm_bf16_128row_tiles"], indent=1))


def sq(d, out):
    c = counters(d)
    rows = []
    for k, v in c.items():
        s = {n: sum(x) for n, x in v.items()}
        r = {"kernel": k, "dispatches": max(len(x) for x in v.values()), **{n: int(t) for n, t in s.items()}}
        if s.get("SQ_BUSY_CU_CYCLES"):
            r["mfma_busy_frac"] = round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * s["SQ_BUSY_CU_CYCLES"]), 4)
        if s.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_frac"] = round(s.get("SQ_LDS_BANK_CONFLICT", 0) / s["SQ_LDS_IDX_ACTIVE"], 4)
        rows.append(r)
    rows.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", 0))
    json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                          "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES -- python bench.py --steps 1 --warmup 1 --cpu-images 0 --concurrent-steps 0",
               "note": "sums over all dispatches of each kernel in the run; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); "
                       "lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "kernels": rows[:16]}, open(out, "w"), indent=1)
    for r in rows[:8]:
        print(r["kernel"][:70], r.get("mfma_busy_frac"), r.get("lds_conflict_frac"))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
    else:
        sq(sys.argv[2], sys.argv[3])
