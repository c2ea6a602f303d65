#!/bin/bash
# tools/rulinalg_variants.sh build | test
#
# rulinalg 0.4.2's utils::dot (src/sound.rs:31) combines its eight running sums either as s + (p0 + p4) or as
# (s + p0) + p4; which one the crate uses could not be checked in this image (include/ssym_rulinalg.h).  The product
# and the oracle take the association from ONE constant; this script builds both with the OTHER value
# (-DSSYM_RULINALG_COMBINE=1) and runs the refcos suite under it, so that either choice is known to be green and
# pinning the crate's is a one-line change.
#   build  (here, no GPU needed): soundsym_amd/csrc/build_rl1/libsoundsym_amd_rl1.so, oracle/libssym_oracle_rl1.so
#   test   (on the GPU box):      the GPU + CPU suites under the variant, minus the committed golden refcos vectors
#                                 (tests/golden/*.npz were generated with the default association)
set -e -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/soundsym_amd/csrc
lib=$csrc/build_rl1/libsoundsym_amd_rl1.so
ora=$root/oracle/libssym_oracle_rl1.so
case "${1:-build}" in
build)
    mkdir -p $csrc/build_rl1
    # only refcos.hip and refcos_mfma.hip see the constant: every other object is the default build's
    make -C $csrc -j4 > /dev/null
    for o in $csrc/build/*.o; do
        b=$(basename $o)
        case $b in refcos.o|refcos_mfma.o) ;; *) cp -p $o $csrc/build_rl1/$b ;; esac
    done
    rm -f $csrc/build_rl1/refcos.o $csrc/build_rl1/refcos_mfma.o
    make -C $csrc -j2 EXTRA=-DSSYM_RULINALG_COMBINE=1 BUILD=build_rl1 OUT=$lib
    make -C $root/oracle -B EXTRA=-DSSYM_RULINALG_COMBINE=1 OUT=$ora
    echo "built $lib and $ora"
    ;;
test)
    export SSYM_LIB=$lib SSYM_ORACLE_LIB=$ora SSYM_RULINALG_COMBINE=1
    cd $root
    python3 -m pytest tests -q -x -m "not gpu" -k "not golden and not needs_the_gpu" -p no:cacheprovider
    python3 -m pytest tests -q -x -m gpu -k "(refcos or chain or topk or api or random or numerics or comm) and not golden and not bench" -p no:cacheprovider
    ;;
*) echo "usage: $0 build|test"; exit 2 ;;
esac
