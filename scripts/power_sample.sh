#!/bin/bash
# Power and clocks of the part while the parity-mode forward runs back to back (run on the GPU box through gpurun):
#   scripts/power_sample.sh [precision] -> rocm-smi samples every 0.5 s during ~300 forwards of 160 slices, then during the fit kernel's bench loop
# What it answers: is the forward running against the board's power limit (the MFMA clock of 1.5-2.0 GHz under load, 2.4 GHz nominal)?
PREC=${1:-fp16x3}
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -4
echo "== idle"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -4
python scripts/prof_unet.py --precision $PREC --slices 160 --batch 160 --reps 300 > /tmp/unet_loop.txt 2>&1 &
PID=$!
sleep 6   # (import + engine set-up)
echo "== during the forward loop ($PREC)"
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //' | paste -sd' ' -
  sleep 0.5
done
tail -c 300 /tmp/unet_loop.txt | cut -c1-200
# ---- the headline fit kernel (fp64 vector work): 500 steps of the bench loop ----
python bench.py --steps 500 --warmup 3 --no-cpu-baseline --no-cfg5 --no-parity --no-unet --recipes A > /tmp/fit_loop.txt 2>&1 &
PID=$!
sleep 8
echo "== during the fit kernel's bench loop (monoexp_lm_kernel, 512 x 512 x 160 x 8 echoes per step)"
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //' | paste -sd' ' -
  sleep 0.5
done
python - <<'PY'
import json
d = json.loads(open("/tmp/fit_loop.txt").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "voxel-fits/s", d["value"])
PY
