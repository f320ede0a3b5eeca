#!/bin/bash
# Address-translation counters of the bench kernels (one rocprofv3 --pmc pass, kernel-trace only): UTCL1 requests / hits / misses per kernel.
#   bash tools/pmc_tlb.sh <tag> [bench.py args...]  -> gpurun_out/<tag>_tlb.txt
export HPF_ENV_SWITCHES=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
ARGS="--steps 5 --warmup 2 --repeats 1 --cpu-iters 0 --no-finish --no-probe --no-single --sweep-1gpu 0 $@"
d=gpurun_out/${TAG}_tlb; rm -rf $d; mkdir -p $d
i=0
for c in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"; do
  mkdir -p $d/g$i
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/g$i -- python3 bench.py $ARGS > $d/g$i/run.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python3 - "$d" > gpurun_out/${TAG}_tlb.txt <<'PY'
import collections, csv, glob, os, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "").replace(" ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r.get("Dispatch_Id"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("TCP_UTCL1_REQUEST_sum", 0)):
    print("%-28s dispatches %4d  " % (k, len(disp[k])) + "  ".join("%s %.4g" % (n.replace("TCP_UTCL1_", "").replace("_sum", ""), v) for n, v in sorted(c.items())))
PY
rm -rf $d
cat gpurun_out/${TAG}_tlb.txt
