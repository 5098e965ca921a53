#!/bin/bash
# round 6, GPU call AF: bit-identity of the commit's weight-gradient kernel with the one of commit beda521, default and float32 modes
O=gpurun_out/r06af; mkdir -p $O
A=$PWD/s-volsdf_amd/lib_ab
python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_new.txt; SVS_LIB_PATH=$A/libsvolsdf_hip_head.so python tools/dev/det_hash.py 4 2>&1 | grep "^dtu\|^bmvs" > $O/hash_old.txt
if cmp -s $O/hash_new.txt $O/hash_old.txt && [ -s $O/hash_new.txt ]; then echo "default precision: BIT-IDENTICAL ($(wc -l < $O/hash_new.txt) lines)"; else echo "default precision: DIFFERENT"; diff $O/hash_new.txt $O/hash_old.txt | head -4; fi | tee $O/identity.txt
SVS_MLP_PRECISION=garbage python tools/dev/det_hash.py 1 2>&1 | tail -2
