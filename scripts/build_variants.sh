set -e
cd "$(dirname "$0")/../multigridcmt_amd/csrc"
rm -rf ../../build/variants ../../variants; mkdir -p ../../build/variants ../../variants
build() { name=$1; shift; make -s -j8 OUT=$PWD/../../variants/lib_$name.so OBJDIR=$PWD/../../build/variants/obj_$name "$@"; echo built $name; }
build nine_rec20 EXTRA=-DMGCMT_RECOMPUTE_NINE_LOG2=20
build nine_rec18 EXTRA=-DMGCMT_RECOMPUTE_NINE_LOG2=18
