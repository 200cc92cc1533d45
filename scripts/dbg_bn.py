import torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from creste_public_amd import train_ops as T
torch.manual_seed(0)
C=16; N,H,W=3,9,14
x=torch.randn(N,C,H,W)*2+0.5; xd=torch.randn(N,C,H,W); gy=torch.randn(N,C,H,W); G=torch.randn(N,C,H,W)
bn=torch.nn.BatchNorm2d(C)
with torch.no_grad(): bn.weight.uniform_(0.5,1.5); bn.bias.normal_(0,0.3)
bn=bn.cuda().train(); gam=bn.weight.detach().cpu().double().view(1,-1,1,1)
op=T.BNT(bn,False)
op.fwd(T.as_act(x.cuda())); op.tan(T.as_act(xd.cuda()))
grads={}
# replicate bwd but keep mom_b
gx,gxd=op.bwd(T.as_act(gy.cuda()),T.as_act(G.cuda()),grads)
X=x.double(); m=lambda z: z.mean(dim=(0,2,3),keepdim=True)
mu=m(X); v=m((X-mu)**2); s=(v+bn.eps)**-0.5; xh=(X-mu)*s
a=xd.double()-m(xd.double()); c=m(xh*xd.double()); t=a-xh*c
print("mean err", (op.mean.cpu().double()-mu.flatten()).abs().max().item(), "invstd err", (op.invstd.cpu().double()-s.flatten()).abs().max().item())
print("mom_t err", (op.mom_t[0].cpu().double()-m(xd.double()).flatten()).abs().max().item(), (op.mom_t[1].cpu().double()-c.flatten()).abs().max().item())
P=lambda z: z-m(z)-xh*m(xh*z)
Gd=G.double(); gyd=gy.double()
gx1=gam*s*P(gyd); gx2=-gam*s*s*(m(Gd*t)*xh+c*P(Gd)+m(Gd*xh)*t)
print("gx err", (gx.nchw().cpu().double()-(gx1+gx2)).abs().max().item(), "gx1-only err", (gx.nchw().cpu().double()-gx1).abs().max().item())
print("gxd err", (gxd.nchw().cpu().double()-gam*s*P(Gd)).abs().max().item())
