#!/bin/bash
# builds the diagnostic probes next to their sources (binaries are git-ignored)
set -e
cd "$(dirname "$0")"
PKG="../../image-captioning-with-external-knowledge_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result probe_small.hip -o probe_small
/opt/rocm/bin/hipcc -O3 -std=c++17 -Wno-unused-result -I ../../include probe_ops.cpp -o probe_ops -L "$PKG" -lick_amd -Wl,-rpath,'$ORIGIN/../../image-captioning-with-external-knowledge_amd'
