#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): builds libick_amd.so + probes there and packs the objects so that the
# caller can unpack them into its own tree (the CPU container this was developed in compiles ~10x slower).
#   gpurun -- 'bash tools/remote_build.sh && <tests...>'   then locally:   tar xzf gpurun_out/build.tgz
set -e
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
bash tools/probes/build_probes.sh > gpurun_out/build_probes.log 2>&1 || { grep -E "error" gpurun_out/build_probes.log; exit 1; }
tar czf gpurun_out/build.tgz image-captioning-with-external-knowledge_amd/libick_amd.so image-captioning-with-external-knowledge_amd/csrc/*.o \
    image-captioning-with-external-knowledge_amd/csrc/.flags tools/probes/probe_ops tools/probes/probe_small oracle/_ref 2>/dev/null || \
tar czf gpurun_out/build.tgz image-captioning-with-external-knowledge_amd/libick_amd.so image-captioning-with-external-knowledge_amd/csrc/*.o \
    image-captioning-with-external-knowledge_amd/csrc/.flags tools/probes/probe_ops tools/probes/probe_small
echo "remote build ok"
