import torch, time
dev='cuda'
for (M,N,K) in [(131072,1024,1024),(131072,1024,1536),(262144,256,256),(262144,256,512)]:
    a=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    b=torch.randn(N,device=dev,dtype=torch.bfloat16)
    for name,fn in [('linear',lambda: torch.nn.functional.linear(a,w,b)),('dW', lambda: a.t()@a[:, :N] if K>=N else None)]:
        if name=='dW' and K<N: continue
        for _ in range(3): fn()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
        fl=2.0*M*N*K if name=='linear' else 2.0*M*K*N
        print(name,M,N,K,'%.1f us %.0f TFLOP/s'%(dt*1e6, fl/dt/1e12))
