#!/bin/bash
# Run on the GPU box: the three probes behind DESIGN 6.3 / 8 / 10, stdout box-stamped -> gpurun_out/r04_probes/
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/r04_probes
mkdir -p $OUT
stamp() { echo "# $(date -u +%FT%TZ) host $(hostname) $(/opt/rocm/bin/rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' | sed 's/.*: *//')"; /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -2 | sed 's/^/# /'; }
for p in mfma_peak mfma_lds; do
  { stamp; echo "# scripts/probes/$p"; timeout 300 $R/scripts/probes/$p; } > $OUT/$p.txt 2>&1
done
{ stamp; echo "# scripts/duplex_probe.py"; timeout 300 python $R/scripts/duplex_probe.py; } > $OUT/duplex.txt 2>&1
cat $OUT/mfma_peak.txt $OUT/mfma_lds.txt $OUT/duplex.txt
