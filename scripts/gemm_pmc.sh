#!/bin/bash
export MDHIP_EXPERIMENTS=1   # the library reads its experiment variables only behind this gate (csrc/md_options.h)
# SQ counters of one GEMM shape/layout: where do the waves wait?  usage: gemm_pmc.sh TAG M K N LAYOUT(NN|NT|TN) [CFG]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out; rm -rf $out/pmc; export TMPDIR=/tmp   # (a fresh directory per run: the summary below globs it)
[ -n "$5" ] && export MDHIP_GEMM_CFG=$5
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc -- python3 scripts/gemm_one.py $1 $2 $3 $4 > $out/run.log 2>&1
python3 - $out <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/pmc/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']:
            agg[r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in agg.items():
    print(k)
    wc=sum(c['SQ_WAVE_CYCLES'])/len(c['SQ_WAVE_CYCLES'])
    for n,v in sorted(c.items()):
        m=sum(v)/len(v); print("   %-28s %14.0f  %6.1f %% of wave cycles" % (n, m, 100*m/wc))
    # waves resident: wave cycles are quad-cycles summed over the launch's waves; MFMA busy cycles are summed over SIMDs
    mf=sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])/len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    print("   launches %d; MFMA busy per SIMD / cycles a wave is resident (1024 SIMDs; waves per SIMD w): %.1f %% x w" % (len(c['SQ_WAVE_CYCLES']), 100 * (mf / 1024) / (4 * wc / 1024)))
PY
