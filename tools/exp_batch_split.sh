cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/expb; mkdir -p $O
for b in 8 4 2; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace$b -o t --output-format rocpd -- python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --batch $b > $O/trace$b.log 2>&1
  python tools/rocpd_step.py $(find $O/trace$b -name "*.db" | head -1) 12 40 > $O/step_B$b.txt 2>&1
  rm -rf $O/trace$b
  echo "B=$b"; head -4 $O/step_B$b.txt; python - <<PY
import re
t=0;n=0
for l in open("$O/step_B$b.txt"):
    if l.startswith("tile_gemm"):
        p=l.split(); n+=int(p[3]); t+=float(p[4])
print("tile_gemm launches", n, "total us", round(t,1))
PY
done
