#!/bin/bash
# usage: meta.sh <object.o|lib.so> <pattern>  -> kernel resource metadata (VGPRs, LDS, scratch) of kernels matching pattern
O=$1; PAT=$2; D=/tmp/co
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $O $D/fat.bin
python3 - "$D/fat.bin" "$PAT" <<'PY'
import sys,re,subprocess,struct
data=open(sys.argv[1],'rb').read(); pat=sys.argv[2]
# concatenated bundles: split at the magic
magic=b'__CLANG_OFFLOAD_BUNDLE__'
pos=[m.start() for m in re.finditer(magic,data)]
for bi,p0 in enumerate(pos):
    blob=data[p0:(pos[bi+1] if bi+1<len(pos) else len(data))]
    n=struct.unpack_from('<Q',blob,24)[0]; off=32
    for _ in range(n):
        o,s,tl=struct.unpack_from('<QQQ',blob,off); off+=24
        t=blob[off:off+tl].decode(); off+=tl
        if 'gfx950' not in t: continue
        open('/tmp/co/x.co','wb').write(blob[o:o+s])
        txt=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes','/tmp/co/x.co'],capture_output=True,text=True).stdout
        for blk in re.split(r'\n\s+- \.agpr_count',txt)[1:]:
            m=re.search(r'\.name:\s+(\S+)',blk)
            if not m or pat not in m.group(1): continue
            g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
            print(m.group(1)[:70],'vgpr',g('vgpr_count'),'agpr',blk.split()[0].strip(': '),'sgpr',g('sgpr_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
PY
