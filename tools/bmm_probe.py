import torch, time
dev='cuda'
b=32; M=5120; K=256; N=5120
A=torch.randn(b,K,M,device=dev); B=torch.randn(b,K,N,device=dev)
C=torch.empty(b,M,N,device=dev)
def run(): torch.bmm(A.transpose(1,2), B, out=C)
for _ in range(3): run()
torch.cuda.synchronize()
a=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
e.record(); torch.cuda.synchronize()
t=a.elapsed_time(e)/5
print(f'torch.bmm (rocBLAS/hipBLASLt) {t*1e3:.0f} us  {2*b*M*N*K/t/1e9:.1f} TFLOP/s')
