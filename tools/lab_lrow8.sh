#!/bin/bash
# TIMING EXPERIMENT for VERDICT r4 item 4a (one-byte row ids in pass 2 of the two-pass pair): a variant library whose pass 2 reads 8
# bytes of row ids per lane and step instead of 16 and decodes nothing (results are WRONG: an upper bound of the gain), against the
# product library, alternating, on config 2.  Build the variant HERE first (hipcc cross-compiles):  bash tools/lab_lrow8.sh build
# then on the GPU box:  bash tools/lab_lrow8.sh
set -u
V=libfastsparse_amd/build/variants
if [ "${1:-}" = "build" ]; then
  mkdir -p $V
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -DFS_LAB_LROW8 -Iinclude \
    -Ilibfastsparse_amd/csrc -c libfastsparse_amd/csrc/fs_kernels_twopass.hip -o $V/fs_kernels_twopass.lrow8.o || exit 1
  objs=$(ls libfastsparse_amd/build/*.o | grep -v fs_kernels_twopass)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $V/fs_kernels_twopass.lrow8.o -o $V/libfs_lrow8.so -lm -ldl -pthread -Wl,-rpath,/opt/rocm/lib || exit 1
  ls -la $V/libfs_lrow8.so
  exit 0
fi
for v in product lrow8 product lrow8; do
  lib=$PWD/libfastsparse_amd/libfastsparse_hip.so
  [ $v = lrow8 ] && lib=$PWD/$V/libfs_lrow8.so
  FS_LIB_PATH=$lib python tools/c2_product_ms.py $v || echo FAIL $v
done
