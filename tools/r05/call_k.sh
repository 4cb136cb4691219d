#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python tools/phase_profile.py --specialized > $O/phase_spec.txt 2>&1
timeout 300 python tools/phase_profile.py --specialized --step-mode 1 > $O/phase_spec_m1.txt 2>&1
grep -v amdgpu.ids $O/phase_spec.txt | tail -24; grep -v amdgpu.ids $O/phase_spec_m1.txt | tail -24
