#!/bin/bash
# rowchain kernel: full / no weight loads (ICK_RC_DBG=1) / no MFMA (ICK_RC_DBG=2)
for d in 0 1 2 3; do echo "== ICK_RC_DBG=$d"; ICK_RC_DBG=$d timeout -k 10 100 tools/probes/probe_ops 200 2>&1 | grep -E "^chain"; done
