#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME "<extra hipcc flags>" file1.hip [file2.hip ...]
# compiles the named translation units with the extra flags and links them with the other objects of the regular build
# into btsbot_amd/libbtsbot_hip_NAME.so (BTSBOT_AMD_LIB=btsbot_amd/libbtsbot_hip_NAME.so python tools/ktime.py).
set -e
cd "$(dirname "$0")/../btsbot_amd/csrc"
name=$1; flags=$2; shift 2
make -s -j8 >/dev/null
mkdir -p /tmp/variant_$name
objs=""
for o in *.o; do
  src=${o%.o}.hip
  hit=0
  for f in "$@"; do [ "$f" == "$src" ] && hit=1; done
  if [ $hit == 1 ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function $flags -c $src -o /tmp/variant_$name/$o
    objs="$objs /tmp/variant_$name/$o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbtsbot_hip_$name.so $objs
echo built ../libbtsbot_hip_$name.so
