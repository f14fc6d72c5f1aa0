import subprocess,os,re,sys
objs=[o for o in os.popen('ls /root/repo/implicit-depth_amd/_obj/*.o').read().split() if not o.endswith('feature_volume.o')]
def build(name,s,defs=()):
    f='/root/repo/implicit-depth_amd/csrc/_pa.hip'
    open(f,'w').write(s)
    r=subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-ffp-contract=off','-fvisibility=hidden','-Wno-unused-function',*defs,'-Rpass-analysis=kernel-resource-usage','-c',f,'-o','/tmp/fv32_%s.o'%name],capture_output=True,text=True)
    os.remove(f)
    if r.returncode: print(name,'FAILED',r.stderr[-600:]); return
    names=re.findall(r'Function Name: (\S+)',r.stderr); sc=re.findall(r'ScratchSize \[bytes/lane\]: (\d+)',r.stderr)
    for n,a in zip(names,sc):
        if 'fv_mlp_kILi' in n: print(name,n[:22],'scratch',a)
    subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-shared','-fPIC']+objs+['/tmp/fv32_%s.o'%name,'-o','/root/repo/implicit-depth_amd/_obj/abl/libidh_ablfv32_%s.so'%name])
cur=open('/root/repo/implicit-depth_amd/csrc/feature_volume.hip').read()
i=cur.index("template <int KT>\n__global__ __launch_bounds__(512) void fv_mlp_k(const FvArgs a) {")
j=cur.index("// Generic variant: any source-view count")
body=cur[i:j]
L1='''            constexpr int KU = KT > 0 ? KT : kMaxK;'''
META='''            const float m3 = P.m3, m4 = P.m4, m5 = P.m5, m6 = P.m6, m10 = P.m10'''
L2='''#pragma unroll
            for (int c = 0; c < kNS; ++c) {
#pragma unroll
                for (int i = 0; i < kNS; ++i) {
#ifdef IDH_ABL_FV2_NOLDSA'''
L3='''            float s = 0.f;
#ifdef IDH_ABL_FV2_NOL3'''
PROEND='''                    wcn = weights(Pn, 0);
                }'''
for x in (L1,META,L2,L3,PROEND): assert body.count(x)==1,x
sp=lambda n:"            __builtin_amdgcn_s_setprio(%d);\n"%n
def var(name, **at):
    b=body
    for key,txt in (('L1',L1),('META',META),('L2',L2),('L3',L3)):
        if key in at: b=b.replace(txt, sp(at[key])+txt)
    if 'PROEND' in at: b=b.replace(PROEND, PROEND+"\n"+sp(at['PROEND']))
    build(name,cur[:i]+b+cur[j:])
var('P_A', L1=1, L2=0)                   # level 1 instead of 2
var('P_C', L1=2, META=0)                 # drop before the metadata MFMAs
var('P_D', L1=2, L2=0, L3=2)             # high in layers 1 and 3
var('P_E', L1=2, L2=1, PROEND=0, L3=2)   # L1 high, prologue chunk middle, rest of L2 low, L3 high
var('P_F', L1=3, L2=0, L3=1)
var('P_G', L1=2, L2=2, PROEND=0)         # high through the next plane's prologue, low for the rest of layer 2
