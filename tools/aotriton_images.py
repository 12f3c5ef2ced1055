"""List / extract the gfx950 code objects of AOTriton's flash-attention kernels that ship inside torch
(torch/lib/aotriton.images/amd-gfx950/flash/<kernel>/*.aks2): an .aks2 file is "AKS2", three u32 (uncompressed size, number of
images, directory bytes) and one LZMA stream holding a directory (LDS bytes, workgroup size, offset, size, name) and the images --
one code object per tuned configuration (BLOCK_M / BLOCK_N / waves ...).  Used in round 6 to read off the arithmetic that decides the
library attention's bits (csrc/ar_attn_exact.hip); which configuration torch launches for a problem is identified from the launch's
grid / workgroup / scratch sizes (tools/gpu/r06_aotriton_configs.py).

  python tools/aotriton_images.py attn_fwd 'bf16@16_128_F_F_0_1' [outdir]      # list (and extract) the images of one functional
  llvm-objdump -d outdir/<image>.hsaco ; llvm-readelf --notes outdir/<image>.hsaco
"""
import os,struct,lzma,sys
ROOT='/usr/local/lib/python3.10/dist-packages/torch/lib/aotriton.images/amd-gfx950/flash'
def load(path):
    b=open(path,'rb').read()
    assert b[:4]==b'AKS2'
    total,n,dirsz=struct.unpack('<III',b[4:16])
    raw=lzma.decompress(b[16:])
    ents=[];p=0
    for i in range(n):
        lds,blk,off,sz,nl=struct.unpack('<IIIII',raw[p:p+20]);p+=20
        name=raw[p:p+nl].rstrip(b'\0').decode();p+=nl
        # skip NUL padding
        ents.append((name,lds,blk,off,sz))
    return ents,raw[dirsz:]
if __name__=='__main__':
    kern,pat=sys.argv[1],sys.argv[2]
    d=os.path.join(ROOT,kern)
    for f in sorted(os.listdir(d)):
        if pat in f:
            ents,blob=load(os.path.join(d,f))
            print(f)
            for e in ents: print('   ',e)
            if len(sys.argv)>3:
                out=sys.argv[3];os.makedirs(out,exist_ok=True)
                for (name,lds,blk,off,sz) in ents:
                    nm=name.split('__P__')[1].split('--')[0]
                    open(os.path.join(out,kern+'__'+nm+'.hsaco'),'wb').write(blob[off:off+sz])
