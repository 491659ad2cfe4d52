#!/bin/bash
# instruction-cache counters of the raster kernel (run on the GPU box): bash tools/dbg/icache_pass.sh [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/icache; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $out/a -o a -- python bench.py --quick --steps 20 --warmup 3 "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/b -o b -- python bench.py --quick --steps 20 --warmup 3 "$@" > $out/b.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "raster" in k or "copy_window" in k:
            print(k, {c: "%.3g" % (v / max(1, cnt[(k, c)])) for c, v in d.items()})
PY
