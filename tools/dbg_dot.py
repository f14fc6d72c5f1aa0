import sys; sys.path.insert(0,'/root/repo')
import torch
import implicit_depth_amd.synthetic as syn
from implicit_depth_amd import _lib
from implicit_depth_amd.cost_volume import CostVolumeManager, to_nhwc, volume_opts
B,K,C,H,W,D=2,3,16,40,80,8
inp={k:v.cuda() for k,v in syn.cost_volume_inputs(B,K,C,H,W,3,2,-1).items()}
m=CostVolumeManager(H,W,D).cuda()
rel=lambda a,b: float((a-b).abs().max()/b.abs().max())
m.kernel=2; ref=m(**inp)[0]
for kern in (2,3):
    m.kernel=kern
    a=m(**inp)[0]
    b=m(**dict(inp,min_depth=0.25,max_depth=5.0))[0]
    print("kern",kern,"device-planes vs quad",rel(a,ref),"float-planes vs quad",rel(b,ref))
    cur=to_nhwc(inp["cur_feats"]); src=to_nhwc(inp["src_feats"])
    out=torch.zeros(B,H,W,D+8,device="cuda"); low=torch.empty(B,H,W,device="cuda"); pl=torch.empty(D,device="cuda")
    opts,_=volume_opts(B,K,C,H,W,D,None,0,0,kernel=kern)
    _lib.check(_lib.lib().idh_cost_volume_dot_ex_fwd(cur.data_ptr(),src.data_ptr(),inp["src_Ks"].data_ptr(),inp["src_extrinsics"].data_ptr(),inp["cur_invK"].data_ptr(),0.25,5.0,B,K,C,H,W,D,out.data_ptr(),D+8,low.data_ptr(),pl.data_ptr(),opts,_lib.stream_ptr()),"x")
    print("   nhwc dense vs quad", rel(out[...,:D].permute(0,3,1,2),ref))
    feats=torch.cat([cur[:,None],src],1).contiguous(); hw=H*W*C
    out.zero_()
    opts,_=volume_opts(B,K,C,H,W,D,None,(K+1)*hw,(K+1)*hw,kernel=kern)
    _lib.check(_lib.lib().idh_cost_volume_dot_ex_fwd(feats.data_ptr(),feats.data_ptr()+4*hw,inp["src_Ks"].data_ptr(),inp["src_extrinsics"].data_ptr(),inp["cur_invK"].data_ptr(),0.25,5.0,B,K,C,H,W,D,out.data_ptr(),D+8,low.data_ptr(),pl.data_ptr(),opts,_lib.stream_ptr()),"x")
    print("   nhwc strided vs quad", rel(out[...,:D].permute(0,3,1,2),ref))
    o2=torch.empty(B,D,H,W,device="cuda")
    _lib.check(_lib.lib().idh_cost_volume_dot_ex_fwd(feats.data_ptr(),feats.data_ptr()+4*hw,inp["src_Ks"].data_ptr(),inp["src_extrinsics"].data_ptr(),inp["cur_invK"].data_ptr(),0.25,5.0,B,K,C,H,W,D,o2.data_ptr(),0,low.data_ptr(),pl.data_ptr(),opts,_lib.stream_ptr()),"x")
    print("   bdhw strided vs quad", rel(o2,ref))
