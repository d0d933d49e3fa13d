import sys, subprocess
S='pp_bf3_fwd-hip-amdgcn-amd-amdhsa-gfx950.s'
lines=open(S).read().split('\n')
BASE=582783  # k77.s line N == file line BASE+N (1-based)
def build(name, inserts=None, replace=None):
    L=list(lines)
    # apply from bottom to top
    for n,txt in sorted((inserts or {}).items(), reverse=True):
        idx=BASE+n-1   # 0-based index of k77 line n
        L.insert(idx+1, txt)
    for n,txt in (replace or {}).items():
        L[BASE+n-1]=txt
    open(name+'.s','w').write('\n'.join(L))
    LLVM='/opt/rocm/lib/llvm/bin/'
    subprocess.check_call([LLVM+'clang','-x','assembler','-target','amdgcn-amd-amdhsa','-mcpu=gfx950','-c',name+'.s','-o',name+'.o'])
    subprocess.check_call([LLVM+'ld.lld','-shared',name+'.o','-o','/root/repo/tools/hsaco/'+name+'.hsaco'])
    print('built',name)
# sanity: show the lines we patch around
for n in (1016,1017,1018,1034,1035,1063,1064):
    print(n, lines[BASE+n-1])
build('orig')
build('pA_nop_between_pk', {1017:'\ts_nop 0'})
build('pB_nop_after_vcmp', {1034:'\ts_nop 3'})
build('pC_nop_after_dswrite', {1063:'\ts_nop 1'})
build('pD_nop_before_pk', {1016:'\ts_nop 1'})
