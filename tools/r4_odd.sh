#!/bin/bash
# round 4: bench.py on configurations other than the headline one (odd batch, other tile sizes, fp32 mode, Swin at another size): must run and print a line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4odd; mkdir -p $O; cd $R
for a in "--batch 3 --size 256" "--batch 2 --size 640" "--batch 5 --size 384 --dtype f32" "--model swin_pop --batch 3 --size 448" "--backbone resnet101 --batch 7 --size 320" "--batch 1 --size 512"; do
  echo "== bench.py $a" >> $O/odd.txt
  timeout 600 python bench.py $a --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs 2>$O/err.txt | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], d['config'])" >> $O/odd.txt 2>&1 || tail -5 $O/err.txt >> $O/odd.txt
done
