export MDHIP_EXPERIMENTS=1   # the library reads its experiment variables only behind this gate (csrc/md_options.h)
cd /tmp; export TMPDIR=/tmp
for b in 128 256 512; do
  MDHIP_ARG_BLOCKS=$b MISC_ONLY=0,10 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_arg_$b -- python3 $GRAFT_REPO_ROOT/scripts/misc_kernels.py > /dev/null 2>&1
  echo "ARG_BLOCKS=$b"; python3 - <<EOF2
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_arg_$b/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]; n=n.split("(anonymous namespace)::")[1] if "(anonymous namespace)::" in n else n
    print("   %-70s calls %4s avg %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3))
EOF2
done
