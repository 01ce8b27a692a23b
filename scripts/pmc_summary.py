"""Summarise rocprofv3 --pmc CSV passes (scripts/pmc_run.sh) for the k_step kernel.

    python scripts/pmc_summary.py gpurun_out/pmc_<tag> [note...] > profiles/<name>.txt
"""
import collections, csv, glob, os, statistics, sys

root = sys.argv[1]
KERNEL = os.environ.get("ANM_PMC_KERNEL", "k_step")  # substring of the kernel name to summarise
print("# rocprofv3 PMC summary of %s (kernel %s, per launch, mean over launches)" % (root, KERNEL))
if len(sys.argv) > 2:
    print("# " + " ".join(sys.argv[2:]))
vals = {}
for p in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(p):
        continue
    f = glob.glob(os.path.join(p, "*", "*_counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = glob.glob(os.path.join(p, "*", "*_kernel_trace.csv"))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if KERNEL in r["Kernel_Name"]]
    print("pass %s: %d launches, avg kernel duration %.2f us" % (os.path.basename(p), len(d), statistics.mean(d)))
    for k, v in sorted(agg.items()):
        vals[k] = statistics.mean(v)
        print("    %-22s %14.4g" % (k, vals[k]))
if "SQ_WAVES" in vals:
    w = vals["SQ_WAVES"]
    print("derived:")
    print("    VALU instructions per wave            %.0f" % (vals["SQ_INSTS_VALU"] / w))
    if "SQ_ACTIVE_INST_VALU" in vals:
        print("    VALU-active / wave cycles             %.1f %%" % (100 * vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"]))
        print("    waiting on s_waitcnt / wave cycles    %.1f %%" % (100 * vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]))
        print("    issue stalls / wave cycles            %.1f %%" % (100 * vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"]))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # rocprofv3 reports KiB; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams by 2x on
    # gfx950 -> doubled (upper bound for this kernel's 8-16 B accesses); WRITE_SIZE is uncalibrated.
    fb, wb = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    print("    FETCH_SIZE  %.2f MB raw, %.2f MB with the gfx950 x2 correction" % (fb / 1e6, 2 * fb / 1e6))
    print("    WRITE_SIZE  %.2f MB raw" % (wb / 1e6))
    print("    HBM traffic per launch (corrected fetch + write) %.2f MB" % ((2 * fb + wb) / 1e6))
    print("    (headline: algorithmic 65536 x 250 B = 16.38 MB; config 4: 16384 x bench.py's algorithmic_bytes_per_transition)")
