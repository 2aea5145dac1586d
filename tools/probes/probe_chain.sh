#!/bin/bash
# row-chain kernels alone (16-wave and 8-wave forms)
timeout -k 10 100 tools/probes/probe_ops 200 2>&1 | grep -E "^chain"
