#!/bin/bash
# phase clock of the fused blocks from the EXPERIMENTS build (tools/ab/libbirda_hip_x.so, built with
#   make -C birda_amd/csrc EXPERIMENTS=1 BUILD=_build_x LIB=../../tools/ab/libbirda_hip_x.so), swapped in for one run
cd ${GRAFT_REPO_ROOT:-.}
cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so
cp tools/ab/libbirda_hip_x.so birda_amd/libbirda_hip.so
python tools/gpu_mb_stamps.py ${1:-1000}
cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so
