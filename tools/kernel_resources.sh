#!/bin/bash
# per-kernel VGPR / AGPR / scratch / LDS of the built engine (reads the gfx950 code object inside librabe_hip.so)
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$(dirname "$0")/../rabe_amd/librabe_hip.so" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o | python3 -c "
import sys, re
for b in sys.stdin.read().split('- .agpr_count')[1:]:
    g = lambda k: (re.search(k + r':\s+(\S+)', b) or [None, '?'])[1]
    print('%-28s agpr %-4s vgpr %-4s scratch %-6s lds %s' % (re.sub(r'^_Z\d+', '', g(r'\.name'))[:28], b.split()[1], g(r'\.vgpr_count'), g(r'\.private_segment_fixed_size'), g(r'\.group_segment_fixed_size')))
"
rm -rf $T
