#!/usr/bin/env python3
"""Copies the judged artefacts of a tools/profile.sh run into profiles/ and updates
profiles/traffic.json (per-launch HBM bytes of the dominant kernel from the PMC passes).

    tools/collect_profile.py gpurun_out/prof_TAG rNN [M nf] [tm] [primary]

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters).  On gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced stream (MI355X_MICROARCH.md, section HBM): the
staging loads of k_run256/k_tile256 are exactly that pattern, so reads are doubled; WRITE_SIZE is
taken as reported (uncalibrated)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
M = int(sys.argv[3]) if len(sys.argv) > 3 else 256
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 262144
# explicit markers instead of guessing from the tag: "tm" = the run's k_run256v2<CF32> launches wrote the tile-major plane
# of the AGC route (their own traffic.json key); "primary" = this run is the kernel's own configuration and may overwrite its key
FLAGS = set(sys.argv[5:])
TILE_MAJOR, PRIMARY = "tm" in FLAGS, "primary" in FLAGS
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_sources_sha16
SRC_SHA = kernel_sources_sha16()
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

summary = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), src], capture_output=True, text=True).stdout
bench_line = ""
tl = os.path.join(src, "trace.log")
if os.path.exists(tl):
    for line in open(tl):
        if line.startswith('{"metric"'):
            bench_line = line.strip()
with open(os.path.join(dst, f"{tag}_rocprofv3_summary.txt"), "w") as f:
    args = open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else "--steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant"
    f.write(f"# rocprofv3 --kernel-trace --stats / --pmc passes of: python3 bench.py {args}   (tools/profile.sh, tools/profile_lite.sh)\n"
            f"# bench line of the traced run:\n# {bench_line}\n\n")
    f.write(summary)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))

vals = defaultdict(dict)
for sub in ("pmc3", "pmc4"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            v = sorted(v)
            vals[k][c] = v[len(v) // 2]                 # median: the first launch after create is not representative (AGC repairs)
tj_path = os.path.join(dst, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}


def short_name(k):
    """rocprofv3's demangled kernel name -> the name csdr_chain_kernel_time reports (bench.py's roofline.kernel)"""
    k = k.replace("void ", "").replace("csdr::(anonymous namespace)::", "").replace("csdr::", "").split("(")[0]
    base, _, targs = k.partition("<")
    first = targs.split(",")[0].strip(" >")
    if first in ("true", "false") and (base.startswith("k_run") or base.startswith("k_tile") or base.startswith("k_shard")):
        # interleaved-shard instantiations carry their stride: k_run256v2<true, 8> -> k_run256v2<FM>/G8 (as csdr_chain_kernel_time names them)
        second = targs.split(",")[1].strip(" >") if "," in targs else ""
        stride = f"/G{second}" if (second.isdigit() and int(second) > 1 and base in ("k_run256v2", "k_run1024v2", "k_shard1024")) else ""
        return base + ("<FM>" if first == "true" else "<CF32>") + stride
    if base == "k_pfb1024" and "," in targs and targs.split(",")[1].strip(" >") == "true":
        return "k_run1024" + ("<FM>" if first == "true" else "<CF32>")     # k_pfb1024<FM, DC = true> is what the C side calls k_run1024
    return base


for k, c in vals.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and "k_" in k:
        rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
        if rd + wr < (1 << 20):
            continue                                    # fix-up / init kernels
        # the AGC configuration's channelizer launch writes its CF32 plane TILE-MAJOR (round 4): its own key
        key = f"{short_name(k)}|M={M}|nf={nf}" + ("|tm" if (TILE_MAJOR and short_name(k) == "k_run256v2<CF32>") else "")
        old = tj.get(key)
        if old and old.get("src_sha16") == SRC_SHA and old.get("source") != f"profiles/{tag}_rocprofv3_summary.txt" and not PRIMARY:
            continue                                    # a kernel's own configuration (PRIMARY=1) wins; other tags do not overwrite it
        tj[key] = {
            "src_sha16": SRC_SHA,
            "hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
            "fetch_size_kib_raw": c["FETCH_SIZE"], "write_size_kib_raw": c["WRITE_SIZE"],
            "note": "median over the launches of one run; FETCH_SIZE x2 (gfx950 wide-load under-count), WRITE_SIZE as reported", "source": f"profiles/{tag}_rocprofv3_summary.txt"}
# the fused 4096-channel route is two launches per call (k_front4096 -> z -> k_back4096<MODE>): bench.py times them as one step under
# the name the plan reports; their bytes add up
fr = [k for k in vals if "k_front4096" in k and "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k]]
bk = [k for k in vals if "k_back4096" in k and "FETCH_SIZE" in vals[k] and "WRITE_SIZE" in vals[k]]
if fr and bk:
    mode = bk[0].split("k_back4096<")[1].split(">")[0].strip()
    name = {"0": "k_front4096+k_back4096<CF32>", "1": "k_front4096+k_back4096<FM>", "2": "k_front4096+k_back4096<FM,mix>"}.get(mode)
    if name:
        rd = sum(vals[k]["FETCH_SIZE"] for k in (fr[0], bk[0])) * 1024 * 2
        wr = sum(vals[k]["WRITE_SIZE"] for k in (fr[0], bk[0])) * 1024
        tj[f"{name}|M={M}|nf={nf}"] = {
            "src_sha16": SRC_SHA, "hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
            "front_fetch_kib_raw": vals[fr[0]]["FETCH_SIZE"], "front_write_kib_raw": vals[fr[0]]["WRITE_SIZE"],
            "back_fetch_kib_raw": vals[bk[0]]["FETCH_SIZE"], "back_write_kib_raw": vals[bk[0]]["WRITE_SIZE"],
            "note": "front + back launch of one call (median each); FETCH_SIZE x2 (gfx950 wide-load under-count; the front's fetches include its four-fold L2-served re-reads only as far as they miss), WRITE_SIZE as reported",
            "source": f"profiles/{tag}_rocprofv3_summary.txt"}
json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
print(open(os.path.join(dst, f"{tag}_rocprofv3_summary.txt")).read()[:1500])
print(json.dumps(tj, indent=1))
