import numpy as np, torch, time
torch.manual_seed(0)
# realistic layer: input post-ReLU activations, weights kaiming-uniform
Cin=Cout=128; H,W=200,176
x=torch.relu(torch.randn(1,Cin,H,W)*1.0+0.3).float()
w=(torch.rand(Cout,Cin,3,3)*2-1).float()*np.sqrt(6.0/(Cin*9))/1.0
ref=torch.nn.functional.conv2d(x.double(),w.double(),padding=1)
d32=torch.nn.functional.conv2d(x,w,padding=1)
def wino(x,w,m):
    # F(m x m, 3x3) in float32 with float32 accumulation (torch matmul)
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
    a=m+2
    U=torch.einsum('ia,ocab,jb->ijoc',torch.tensor(G),w.double(),torch.tensor(G)).float()   # U computed in f64, stored f32
    xp=torch.nn.functional.pad(x,(1,1,1,1))
    th,tw=H//m,W//m
    # patches (th,tw,C,a,a)
    P=xp.unfold(2,a,m).unfold(3,a,m)[0]  # C,th,tw,a,a
    BTt=torch.tensor(BT).float()
    V=torch.einsum('ia,ctuab,jb->ijctu',BTt,P,BTt)  # f32 transform
    V=V.reshape(a,a,Cin,th*tw)
    M=torch.matmul(U,V)  # (a,a,Cout,T) f32
    ATt=torch.tensor(AT).float()
    Y=torch.einsum('ia,abot,jb->otij',ATt,M,ATt)  # Cout,T,m,m
    Y=Y.reshape(Cout,th,tw,m,m).permute(0,1,3,2,4).reshape(1,Cout,H,W)
    return Y
mx=float(ref.abs().max())
print('max|ref|',mx,'mean|ref|',float(ref.abs().mean()))
for name,y in (('direct f32',d32),('F(2x2) f32',wino(x,w,2)),('F(4x4) f32',wino(x,w,4))):
    e=(y.double()-ref).abs()
    print('%-12s max err %.3e  (rel max %.2e)  mean err %.3e'%(name,float(e.max()),float(e.max())/mx,float(e.mean())))
