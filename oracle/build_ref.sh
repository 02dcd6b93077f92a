#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.  Builds the reference encoder/decoder
# (HM 16.20 + the reference's `modified2019` edits) from the sources WHERE THEY LIE
# under /root/reference/HM_dl/source into oracle/_ref/ (git-ignored, never copied
# into the repo).  Flags follow HM_dl/build/linux/common/makefile.base:52,76
# (-O3 -std=c++11) plus -ffp-contract=off (no-op on baseline x86-64).
#
# Windows-only lines: TLibEncoder/TEncCu.cpp:44-45 include <io.h>/<Windows.h> solely
# for the label-file polling line :245 (`_access`, `Sleep`) -- the file IPC this
# project replaces, not part of the algorithm.  That one translation unit is fed to
# the compiler through a pipe with those two #include lines dropped and the two
# names mapped to their POSIX equivalents on the command line; no header, library
# or generated file is fabricated, no reference source is modified or copied.
#
# oracle/ref_hook.cpp (ours) is linked in with GNU ld --wrap so that the outputs
# of TEncCu::compressCtu can be observed (see that file).
set -e
REF=${REF:-/root/reference/HM_dl/source}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF" ] || { echo "reference sources not present; keeping prebuilt $OUT"; exit 0; }
mkdir -p $OUT/obj $OUT/objd
CXX="g++ -std=c++11 -O3 -ffp-contract=off -w -DMSYS_LINUX -DEXTENSION_360_VIDEO=0 -I$REF/Lib -I$REF/App/TAppEncoder"
compile() { # src obj
  local src=$1 obj=$2
  [ "$obj" -nt "$src" ] && return 0
  case "$src" in
    */TEncCu.cpp)
      sed -e '/#include *<io.h>/d' -e '/#include *<Windows.h>/d' "$src" | \
        $CXX -x c++ -include unistd.h -D_access=access '-DSleep(ms)=usleep(1000*(ms))' \
             -I"$(dirname "$src")" -c - -o "$obj" ;;
    *.c) gcc -O3 -w -c "$src" -o "$obj" ;;
    *) $CXX -c "$src" -o "$obj" ;;
  esac
}
export -f compile; export CXX REF
ENC_SRCS=$(ls $REF/Lib/TLibCommon/*.cpp $REF/Lib/TLibEncoder/*.cpp $REF/Lib/TLibVideoIO/*.cpp \
              $REF/Lib/TAppCommon/*.cpp $REF/App/TAppEncoder/*.cpp $REF/Lib/libmd5/libmd5.c)
for f in $ENC_SRCS; do echo "$f $OUT/obj/$(basename ${f%.*}).o"; done | xargs -P8 -L1 bash -c 'compile $0 $1'
$CXX -c $HERE/ref_hook.cpp -o $OUT/obj/ref_hook.o
g++ -o $OUT/TAppEncoder_ref $OUT/obj/*.o -lpthread \
    -Wl,--wrap=_ZN6TEncCu11compressCtuEiP10TComDataCU
if [ "$1" = "--decoder" ]; then
  DEC_SRCS=$(ls $REF/Lib/TLibDecoder/*.cpp $REF/App/TAppDecoder/*.cpp)
  for f in $DEC_SRCS; do echo "$f $OUT/objd/$(basename ${f%.*}).o"; done | xargs -P8 -L1 bash -c 'compile $0 $1'
  COMMON=""
  for f in $REF/Lib/TLibCommon/*.cpp $REF/Lib/TLibVideoIO/*.cpp $REF/Lib/TAppCommon/*.cpp $REF/Lib/libmd5/libmd5.c; do
    COMMON="$COMMON $OUT/obj/$(basename ${f%.*}).o"; done
  g++ -o $OUT/TAppDecoder_ref $OUT/objd/*.o $COMMON -lpthread
fi
# The ANCHOR of the BD-rate comparison: the same encoder with the label pruning switched off, i.e. the
# behaviour of unmodified HM 16.20 (the reference ships it only as TAppEncoder_original.exe).  Same pipe technique on the same single
# translation unit: one statement, `check_current = check_next = true;`, is inserted in the stream in front of TEncCu.cpp:522
# (`Int iBaseQP = xComputeQP(`), after the label lookup has set the two flags.  It is an anchor for rate / PSNR, not a parity pin.
SRC=$REF/Lib/TLibEncoder/TEncCu.cpp
if [ ! $OUT/obj_anchor_TEncCu.o -nt $SRC ]; then
  sed -e '/#include *<io.h>/d' -e '/#include *<Windows.h>/d' \
      -e 's/^\( *\)Int iBaseQP = xComputeQP( rpcBestCU, uiDepth );/\1check_current = check_next = true; Int iBaseQP = xComputeQP( rpcBestCU, uiDepth );/' "$SRC" | \
    $CXX -x c++ -include unistd.h -D_access=access '-DSleep(ms)=usleep(1000*(ms))' -I"$(dirname "$SRC")" -c - -o $OUT/obj_anchor_TEncCu.o
fi
g++ -o $OUT/TAppEncoder_anchor $(ls $OUT/obj/*.o | grep -v '/TEncCu.o$') $OUT/obj_anchor_TEncCu.o -lpthread \
    -Wl,--wrap=_ZN6TEncCu11compressCtuEiP10TComDataCU
# STAGE TRACES: the encoder with HM's own two trace switches on -- DEBUG_INTRA_SEARCH_COSTS (TypeDef.h:59: cost lines of the
# mode search, TEncSearch.cpp:2315,2395) and DEBUG_TRANSFORM_AND_QUANTISE (TypeDef.h:60: every TU block on its way through transform, quantiser,
# dequantiser and inverse transform, TComTrQuant.cpp:1496-1658).  TypeDef.h sets both to 0 unconditionally, so the two translation units that test
# them are piped to the compiler with an #undef/#define pair inserted behind their last #include; nothing else differs from TAppEncoder_ref.
tr_unit() { # src macro anchor-include obj
  [ "$4" -nt "$1" ] && return 0
  sed -e "s|^\(#include $3\)\$|\1\n#undef $2\n#define $2 1|" "$1" | $CXX -x c++ -I"$(dirname "$1")" -c - -o "$4"
}
tr_unit $REF/Lib/TLibEncoder/TEncSearch.cpp DEBUG_INTRA_SEARCH_COSTS '<limits>' $OUT/obj_trace_TEncSearch.o
tr_unit $REF/Lib/TLibCommon/TComTrQuant.cpp DEBUG_TRANSFORM_AND_QUANTISE '"Debug.h"' $OUT/obj_trace_TComTrQuant.o
g++ -o $OUT/TAppEncoder_trace $(ls $OUT/obj/*.o | grep -v -e '/TEncSearch.o$' -e '/TComTrQuant.o$') $OUT/obj_trace_TEncSearch.o $OUT/obj_trace_TComTrQuant.o -lpthread \
    -Wl,--wrap=_ZN6TEncCu11compressCtuEiP10TComDataCU
echo "built: $(ls $OUT | grep -v obj | tr '\n' ' ')"
