import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
import test_distributed_gpu as T

def hsh(a):
    a = np.asarray(a)
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:8] + ("F" if a.flags.f_contiguous and not a.flags.c_contiguous else "C") + str(a.ctypes.data % 64)

def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import runia_core_amd as rc
    from runia_core_amd.distributed import broadcast_fitted
    from runia_core_amd.inference import LaREMPipeline, MDLatentSpace
    state = None
    if rank == 0:
        xtr, rtr, ftr = T._inputs(700, 1)
        probe = LaREMPipeline(None, None, T.N_MC, 0.5, 2)
        h_train = probe.entropy(probe.stack(xtr, rtr)).cpu().numpy()
        np.random.seed(3)
        red, pca = rc.apply_pca_ds_split(h_train, T.N_PCA)
        md = MDLatentSpace(); md.setup(red)
        state = {"md": md, "pca": pca}
    state = broadcast_fitted(state, min_tensor_bytes=1024)
    md, pca = state["md"], state["pca"]
    pipe = LaREMPipeline(md, pca, T.N_MC, 0.5, 2)
    x, rand, _ = T._inputs(1001, 1101)
    s = pipe.score_latents(x, rand).cpu().numpy()
    f = pipe._folded_state()
    h = pipe.entropy_from_latents(x, rand).cpu().numpy()
    print(rank, "prec", hsh(md.precision), "mean", hsh(md.feats_mean), "comp", hsh(pca.components_), "pmean", hsh(pca.mean_), "var", hsh(pca.explained_variance_),
          "folded M", hsh(f[0].cpu().numpy()), "c", hsh(f[1].cpu().numpy()), "r", f[2], "h", hsh(h), "scores", hsh(s), "centered", hsh(md.centered_data), flush=True)
    lam, vec = np.linalg.eigh((np.asarray(md.precision) + np.asarray(md.precision).T) * 0.5)
    print(rank, "eigh lam", hsh(lam), "vec", hsh(vec), "bias", hsh(pipe.pca.bias_host), "scale", hsh(pipe.pca.scale_host), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2, T._free_port()), nprocs=2, join=True)
