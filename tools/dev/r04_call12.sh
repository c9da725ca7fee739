#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 100 tools/dev/mfma_hazard_probe pkx > $O/r04_pkx_probe.txt 2>&1; echo "pkx rc $?"; cut -c1-330 $O/r04_pkx_probe.txt
