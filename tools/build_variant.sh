#!/bin/bash
# Builds the library from another git revision into sparkfm_amd/lib/libfmhip_<name>.so (A/B runs on one
# GPU box: FMHIP_LIB=sparkfm_amd/lib/libfmhip_<name>.so python tools/ab_bench.py ...).
#   tools/build_variant.sh <name> [git-rev, default HEAD; WORK = the working tree]
#   EXTRA_FLAGS="-DFOO=1" adds compiler flags (experiment macros)
set -e
name=$1; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/fmhip_variant.XXXXXX)
mkdir -p $tmp/sparkfm_amd/csrc $tmp/include
for f in fm_forward.hip fm_backward.hip fm_apply.hip fm_device.h fm_kernels.h fm_constants.h als_kernels.hip als_kernels.h csc_build.hip csc_build.h fmhip_api.hip fmhip_dataset.hip fmhip_step.hip fmhip_comm.hip fmhip_internal.h fmhip_host.h fmhip_host.cpp; do
  if [ "$rev" = WORK ]; then cp $root/sparkfm_amd/csrc/$f $tmp/sparkfm_amd/csrc/$f; else git -C $root show $rev:sparkfm_amd/csrc/$f > $tmp/sparkfm_amd/csrc/$f; fi
done
for h in fmhip.h fmhip_experimental.h; do
  if [ "$rev" = WORK ]; then cp $root/include/$h $tmp/include/$h; else git -C $root show $rev:include/$h > $tmp/include/$h; fi
done
objs=""
for f in fm_forward.hip fm_backward.hip fm_apply.hip als_kernels.hip csc_build.hip fmhip_api.hip fmhip_dataset.hip fmhip_step.hip fmhip_comm.hip fmhip_host.cpp; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC ${EXTRA_FLAGS:+-DFMHIP_ABLATION_BUILD} $EXTRA_FLAGS -c $tmp/sparkfm_amd/csrc/$f -o $tmp/${f%.*}.o &
  objs="$objs $tmp/${f%.*}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/sparkfm_amd/lib/libfmhip_$name.so $objs -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
rm -rf $tmp
echo $root/sparkfm_amd/lib/libfmhip_$name.so
