import subprocess,re
S='pp_bf3_fwd-hip-amdgcn-amd-amdhsa-gfx950.s'
lines=open(S).read().split('\n')
BASE=582783
def build(name, inserts, vg=None):
    L=list(lines)
    if vg:
        # kernel descriptor of this kernel follows its body
        for i in range(BASE+3000, BASE+6000):
            if '.amdhsa_next_free_vgpr 107' in L[i]: L[i]=L[i].replace('107',str(vg))
            if '.amdhsa_accum_offset 108' in L[i]: L[i]=L[i].replace('108',str(vg)); break
    for n,txt in sorted(inserts.items(), reverse=True):
        L.insert(BASE+n, txt)
    open(name+'.s','w').write('\n'.join(L))
    LLVM='/opt/rocm/lib/llvm/bin/'
    subprocess.check_call([LLVM+'clang','-x','assembler','-target','amdgcn-amd-amdhsa','-mcpu=gfx950','-c',name+'.s','-o',name+'.o'])
    subprocess.check_call([LLVM+'ld.lld','-shared',name+'.o','-o','/root/repo/tools/hsaco/'+name+'.hsaco'])
    print('built',name)
def cnt(reg, off):
    return f'''\tv_cmp_eq_f32_e64 s[2:3], 0, {reg}
\ts_nop 1
\ts_and_b64 exec, s[6:7], s[2:3]
\tglobal_atomic_add v114, v115, s[4:5] offset:{off}'''
post='''\ts_mov_b64 s[6:7], exec
\tv_readlane_b32 s2, v106, 2
\tv_readlane_b32 s3, v106, 3
\ts_nop 4
\ts_load_dwordx2 s[4:5], s[2:3], 0x8
\tv_mov_b32_e32 v114, 0
\tv_mov_b32_e32 v115, 1
\ts_waitcnt lgkmcnt(0)
'''+'\n'.join(cnt(r,o) for r,o in (('v108',0),('v109',4),('v110',8),('v111',12),('v112',16),('v113',20)))+'''
\ts_mov_b64 exec, s[6:7]
\tglobal_atomic_add v114, v115, s[4:5] offset:24'''
ins={1016:'\tv_mov_b32_e32 v111, v92\n\tv_mov_b32_e32 v112, v99\n\tv_mov_b32_e32 v113, v94',
     1017:'\tv_mov_b32_e32 v108, v48',
     1018:'\tv_mov_b32_e32 v109, v48',
     1020:'\tv_mov_b32_e32 v110, v48',
     1066:post}
build('dbg2', ins, vg=116)
# same allocation change but no instrumentation: does the allocation size alone matter?
build('alloc116', {}, vg=116)
