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

ACT="""            f32x4 acc2[kNS];
            Pro Pn;"""
assert body.count(ACT)==1
def var2(name, edit=None, defs=(), **at):
    b=body
    for key,txt in (('L1',L1),('META',META),('ACT',ACT),('L2',L2),('L3',L3)):
        if key in at: b=b.replace(txt, sp(at[key])+txt)
    if edit: b=edit(b)
    build(name,cur[:i]+b+cur[j:],defs)
var2('Q_K', L1=2, ACT=0)
def chunk(n):
    def e(b):
        assert b.count("if (c == (KT > 0 ? 0 : 1)) {")==1
        return b.replace("if (c == (KT > 0 ? 0 : 1)) {","if (c == (KT > 0 ? %d : 1)) {"%n)
    return e
var2('Q_H1', chunk(1), L1=2, L2=0)
var2('Q_H2', chunk(2), L1=2, L2=0)
var2('Q_V2', None, ('-DIDH_ABL_FV_VALU_PER_GROUP=2',), L1=2, L2=0)
var2('Q_V4', None, ('-DIDH_ABL_FV_VALU_PER_GROUP=4',), L1=2, L2=0)
def nosgb(b):
    assert b.count("if constexpr (KT > 0) {\n                    // Scheduling hint")==1
    return b.replace("if constexpr (KT > 0) {\n                    // Scheduling hint","if constexpr (false) {\n                    // Scheduling hint")
var2('Q_NOSGB', nosgb, L1=2, L2=0)
