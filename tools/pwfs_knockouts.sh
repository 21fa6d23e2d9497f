#!/bin/bash
# Knock-out table of the one-pass split-bf16 pointwise backward (csrc/pwfuseds.hip): parts of the kernel switched off through CFN_PWFS_DBG
# (1 weight gradient, 2 data gradient, 4 act' / statistics epilogue, 8 gx stores) -- results are wrong, times tell where a stage goes.  GPU box.
R=${GRAFT_REPO_ROOT:-$PWD}
for d in 0 1 2 3 4 8 7 15; do
  echo "## CFN_PWFS_DBG=$d"
  CFN_PWF_SPLIT=2 CFN_PWFS_DBG=$d python $R/tools/pwfs_bench.py 2>&1 | grep "fused" | sed 's/separate [0-9.]* ms *//'
done
