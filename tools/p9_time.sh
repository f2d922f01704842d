# A/B of the 3x3 patch kernel (SEGLAND_CONV_P9=1) against the half-tile kernel
for v in 1 0; do echo "== SEGLAND_CONV_P9=$v"; for cfg in "--hw 64 --cin 512 --cout 512 --k 3 --dil 4" "--hw 64 --cin 256 --cout 256 --k 3 --dil 2" "--hw 64 --cin 2048 --cout 512 --k 3"; do echo "$cfg"; SEGLAND_CONV_P9=$v python tools/conv_time.py $cfg | grep -v wgrad; done; done
