import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from msa_amd import ops
dev="cuda"; heads=2; H=128
for lens in ([64], [128], [50], [200]):
    M=sum(lens); torch.manual_seed(0)
    qkv=torch.randn(M,3*H).bfloat16(); dctx=torch.randn(M,H).bfloat16(); bias=torch.zeros(M)
    lay=ops.SeqLayout(lens,heads,dev)
    ctx,lse=ops.attn_fwd(qkv.to(dev),bias.to(dev),lay,H)
    dq=ops.attn_bwd(qkv.to(dev),ctx,dctx.to(dev),lse,bias.to(dev),lay,H).float().cpu()
    x=qkv.float().requires_grad_(True); n=lens[0]
    q,k,v=(x[:, j*H:(j+1)*H].view(n,heads,64).transpose(0,1) for j in range(3))
    w=torch.softmax(q@k.transpose(-1,-2)*0.125,-1); out=(w@v).transpose(0,1).reshape(n,H)
    out.backward(dctx.float())
    for j,nm in enumerate(("dq","dk","dv")):
        e=(dq[:, j*H:(j+1)*H]-x.grad[:, j*H:(j+1)*H]).abs()
        rows=(e.max(1).values>0.05).nonzero().flatten().tolist()
        print(lens, nm, "max err", float(e.max()), "bad rows", rows[:12], "n", len(rows))
print("---- poison test")
lens=[64]; M=64
qkv=torch.randn(M,3*H).bfloat16(); dctx=torch.randn(M,H).bfloat16(); bias=torch.zeros(M)
lay=ops.SeqLayout(lens,heads,dev)
ctx,lse=ops.attn_fwd(qkv.to(dev),bias.to(dev),lay,H)
out=torch.full((M,3*H), float('nan'), device=dev, dtype=torch.bfloat16)
ops.attn_bwd(qkv.to(dev),ctx,dctx.to(dev),lse,bias.to(dev),lay,H,dqkv=out)
torch.cuda.synchronize()
o=out.float().cpu()
for j,nm in enumerate(("dq","dk","dv")):
    print(nm, "nan frac", float(o[:, j*H:(j+1)*H].isnan().float().mean()))
print("ftiles", lay.ftile_seq.tolist(), lay.ftile_r0.tolist(), "tiles", lay.tile_seq.tolist(), lay.tile_r0.tolist())
print("---- correlation test")
x=qkv.float(); n=64
q,k,v=(x[:, j*H:(j+1)*H].view(n,heads,64).transpose(0,1) for j in range(3))
dO=dctx.float().view(n,heads,64).transpose(0,1)
sc=q@k.transpose(-1,-2)*0.125
P=torch.softmax(sc,-1)
dv_ref=(P.transpose(-1,-2)@dO).transpose(0,1).reshape(n,H)
dv_unn=(torch.exp(sc).transpose(-1,-2)@dO).transpose(0,1).reshape(n,H)
dv_k=o[:,2*H:]
def corr(a,b): return float((a*b).sum()/a.norm()/b.norm())
print("corr(dv, ref)", corr(dv_k,dv_ref), "corr(dv, unnormalised)", corr(dv_k,dv_unn), "norm ratio", float(dv_k.norm()/dv_ref.norm()))
print("lse kernel", lse[:4].cpu().tolist(), "ref", torch.logsumexp(sc,-1)[:, :4].t().tolist())
# per-row check: which P would reproduce dv? solve least squares on head 0
print("max abs dq,dk,dv:", [float(o[:, j*H:(j+1)*H].abs().max()) for j in range(3)])
print("dq row0[:8]", o[0,:8].tolist())
print("---- candidates")
dP=dO@v.transpose(-1,-2)                      # [h, q, k]
delta=(dP*P).sum(-1,keepdim=True)
def pack(t): return t.transpose(0,1).reshape(n,H)
cands={
 "dq_ref": pack((P*(dP-delta))@k*0.125),
 "dq_nodelta": pack((P*dP)@k*0.125),
 "dq_onlydelta": pack((P*(-delta))@k*0.125),
 "dk_ref": pack((P*(dP-delta)).transpose(-1,-2)@q*0.125),
 "dk_nodelta": pack((P*dP).transpose(-1,-2)@q*0.125),
 "dk_onlydelta": pack((P*(-delta)).transpose(-1,-2)@q*0.125),
}
for nm,c in cands.items():
    kk=o[:, :H] if nm.startswith("dq") else o[:, H:2*H]
    print(nm, "corr", round(corr(kk,c),4), "norm ratio", round(float(kk.norm()/c.norm()),4))
print("---- dscale experiment")
for dr in [(0,0,1.0),(0,0,2.0),(123,0,1.0)]:
    out=torch.zeros((M,3*H), device=dev, dtype=torch.bfloat16)
    ops.attn_bwd(qkv.to(dev),ctx,dctx.to(dev),lse,bias.to(dev),lay,H,dqkv=out,drop=dr)
    oo=out.float().cpu()
    print(dr, "dq corr ref", round(corr(oo[:,:H],cands["dq_ref"]),4), "onlydelta", round(corr(oo[:,:H],cands["dq_onlydelta"]),4), "dv max", float(oo[:,2*H:].abs().max()))
