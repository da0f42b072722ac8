"""Does gloo all-reduce a CUDA tensor on this build?  Two ranks on one GPU; prints progress to a file."""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp

def w(rank, port):
    log = open('gpurun_out/probe_gloo_%d.log' % rank, 'w')
    def say(*a):
        print(*a, file=log, flush=True)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    say('init ok')
    t = torch.full((1024,), float(rank + 1))
    dist.all_reduce(t); say('cpu all_reduce', t[0].item())
    x = torch.full((1024,), float(rank + 1), device='cuda:0')
    say('cuda tensor made')
    work = dist.all_reduce(x, async_op=True); say('issued')
    work.wait(); torch.cuda.synchronize(); say('cuda all_reduce', x[0].item())
    dist.destroy_process_group(); say('done')

if __name__ == '__main__':
    os.makedirs('gpurun_out', exist_ok=True)
    ctx = mp.get_context('spawn')
    ps = [ctx.Process(target=w, args=(r, 29533)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(40)
        if p.is_alive():
            p.kill(); print('rank hung')
    print('exit codes', [p.exitcode for p in ps])
