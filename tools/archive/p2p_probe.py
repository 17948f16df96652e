import os, sys, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.getcwd())
def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kagnn_amd import p2p
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, out = 1000, 16
    b = p2p.PeerBuffers(n * out, dev)
    part = b.local.view(n, out)
    part.copy_(torch.arange(n * out, device=dev, dtype=torch.float32).view(n, out) * (rank + 1))
    p2p.rank_barrier()
    y = p2p.reduce_scatter(b, n, out)
    want = (torch.arange(n * out, dtype=torch.float32).view(n, out) * 3)[:, rank * 8:(rank + 1) * 8]
    print(rank, "rs ok", torch.equal(y.cpu(), want))
    p2p.rank_barrier()
    b.local[: n * 8].view(n, 8).copy_(y)
    p2p.rank_barrier()
    g = p2p.all_gather(b, n, 8)
    print(rank, "ag ok", torch.equal(g.cpu(), torch.arange(n * out, dtype=torch.float32).view(n, out) * 3))
    p2p.rank_barrier()
    dist.destroy_process_group()
if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
