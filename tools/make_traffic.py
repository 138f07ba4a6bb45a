"""Turn the FETCH_SIZE / WRITE_SIZE rocprofv3 passes of tools/prof_bench.sh into profiles/<tag>/traffic.json.
   python tools/make_traffic.py <prof_dir> <out_json> <rows_per_launch>
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2 before comparing with byte counts."""
import csv, glob, json, os, sys
from collections import defaultdict

d, out, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])


def collect(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")


def kernel_avg_ms():
    """Kernel_Name -> average duration (ms) from the --kernel-trace --stats pass of the same command (kernel_stats.csv)."""
    out = {}
    f = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = float(r["AverageNs"]) / 1e6
    return out


avg_ms = kernel_avg_ms()
# kernel-name patterns per class: the LayerNorm-folded chain (default) uses epilogues 6 / 7 / 5, the separate-LayerNorm chain 0 / 1 / 2
CLASSES = {"gemm_qkv": (("gemm_pp6_kernel<T_F16, 6", "gemm_pp6_kernel<T_F16, 0", "gemm_pp_kernel<T_F16, 6,", "gemm_pp_kernel<T_F16, 0,"), 2 * 1024 + 2 * 3072),
           "gemm_fc1": (("gemm_pp6_kernel<T_F16, 7", "gemm_pp6_kernel<T_F16, 1", "gemm_pp_kernel<T_F16, 7,", "gemm_pp_kernel<T_F16, 1,"), 2 * 1024 + 2 * 4096),
           # round 3: fc2 runs the residual epilogue on the 384 x 256 kernel, the out-projection stays on 256 x 256 (both EPI 5);
           # before that the two shared one kernel name ("gemm_out_fc2_mixed")
           "gemm_fc2": (("gemm_pp6_kernel<T_F16, 5",), 2 * 4096 + 4 * 1024 + 4 * 1024 + 2 * 1024),
           "gemm_out": (("gemm_pp_kernel<T_F16, 5,", "gemm_pp_kernel<T_F16, 2,"), 2 * 1024 + 4 * 1024 + 4 * 1024 + 2 * 1024),
           "attention": (("attention8_kernel",), 2 * 3072 + 2 * 1024),          # (attention_f32_kernel: the exact pass, a few images)
           "layernorm": (("layernorm_kernel",), 4 * 1024 + 2 * 1024),
           "refine_candidates": (("refine_candidates_kernel<false>", "refine_candidates_kernel"), None)}    # <true>: the certainty pass
TAIL_SPLIT = ("gemm_fc1", "gemm_fc2")                # the product's tail policy: K >= 2048 or N >= 4096 (vit.hip PG_DEFAULT_GEMM_TAIL_*)
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, one launch = %d token rows; KiB as "
               "reported; hbm_bytes_per_launch_corrected = 2 x FETCH_SIZE (gfx950 under-report of wide coalesced reads, "
               "MI355X_MICROARCH.md) + WRITE_SIZE; algorithmic_bytes = activation rows in + out (+ weights once)" % rows}
for cls, (pats, bytes_per_row) in CLASSES.items():
    f, w = [], []
    for pat in pats:                                   # first pattern with data wins
        f = [v for k, v in fetch.items() if pat in k]
        w = [v for k, v in write.items() if pat in k]
        if f and w:
            break
    if not f or not w:
        continue
    e = {"rows": rows, "fetch_size_kib_raw": f[0], "write_size_kib": w[0],
         "hbm_bytes_per_launch_corrected": (2 * f[0] + w[0]) * 1024}
    for pat in pats:                                   # rocprofv3 average launch duration of the same kernel class
        ms = [v for k, v in avg_ms.items() if pat in k]
        if ms:
            e["rocprof_avg_ms"] = ms[0]
            # ... and of the small-tile launch that takes the rows beyond the persistent kernel's last whole round (gemm_tail.hip,
            # same epilogue number): a GEMM of the model = the two launches together
            epi = pat.split("<T_F16,")[1].strip().split(",")[0].split(">")[0].strip() if "<T_F16," in pat else None
            # (round 6: a K = 1024 tail -- fc1's -- goes through gemm_mid.hip; in a 512-image step that kernel name has no other use)
            tail = [v for k, v in avg_ms.items() if epi is not None and (f"gemm_tail_kernel<T_F16, {epi}>" in k.replace(" >", ">") or
                                                                         f"gemm_mid_kernel<T_F16, {epi}>" in k.replace(" >", ">"))]
            if tail and cls in TAIL_SPLIT:
                e["rocprof_tail_avg_ms"] = tail[0]
            break
    if bytes_per_row:
        wbytes = {"gemm_qkv": 3072 * 1024 * 2, "gemm_fc1": 4096 * 1024 * 2, "gemm_fc2": 1024 * 4096 * 2, "gemm_out": 1024 * 1024 * 2}.get(cls, 0)
        e["algorithmic_bytes"] = rows * bytes_per_row + wbytes
    res[cls] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
