#!/bin/bash
# tools/ab/base.so = the library built from the kernel sources of a git revision (default HEAD), for tools/ab_libs.sh
REV=${1:-HEAD}
D=/root/repo/gpurun_out/base_src
rm -rf $D; mkdir -p $D/pkg/csrc $D/include /root/repo/tools/ab
for f in cutseq_hip.hip trim_kernel.hip.inc long_kernel.hip.inc text_kernels.hip.inc deflate_kernels.hip.inc; do git -C /root/repo show $REV:cutseq_amd/csrc/$f > $D/pkg/csrc/$f; done
git -C /root/repo show $REV:include/cutseq_hip.h > $D/include/cutseq_hip.h
cd $D/pkg/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -o /root/repo/tools/ab/base.so cutseq_hip.hip && echo built tools/ab/base.so from $REV
