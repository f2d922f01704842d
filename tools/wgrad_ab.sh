#!/bin/bash
# A/B of the 256 x 256 weight-gradient kernel: eight waves of 128 x 64 (default) against four waves of 128 x 128 (SEGLAND_WGRAD_W4=1), per bench shape
for cfg in "--hw 64 --cin 2048 --cout 512 --k 3" "--hw 64 --cin 512 --cout 512 --k 3 --dil 4" "--hw 64 --cin 256 --cout 256 --k 3 --dil 2" "--hw 64 --cin 512 --cout 2048" "--hw 64 --cin 2048 --cout 512" "--hw 64 --cin 1024 --cout 2048"; do
  for v in 0 1 0 1; do echo -n "W4=$v $cfg: "; SEGLAND_WGRAD_W4=$v python3 tools/conv_time.py $cfg | grep wgrad; done
done
