import os, sys, socket
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adaflo_amd
from adaflo_amd import parallel
from common import rel_l2

def view(g, part, degree):
    sl = []
    for d in (2, 1, 0):
        lo = part.coords[d] * part.cells[d] * degree
        sl.append(slice(lo, lo + part.cells[d] * degree + 1))
    return g[tuple(sl)]

def mult(part, degree):
    nn = part.nodes(degree)
    m = np.ones((nn[2], nn[1], nn[0], 1))
    for d, axis in ((0, 2), (1, 1), (2, 0)):
        idx = [slice(None)] * 4
        if part.coords[d] > 0:
            idx[axis] = 0; m[tuple(idx)] *= 2
        idx = [slice(None)] * 4
        if part.coords[d] < part.grid[d] - 1:
            idx[axis] = -1; m[tuple(idx)] *= 2
    return m

def worker(rank, world, port, grid, cells):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    k = 2
    part = parallel.BrickPartition(grid, rank, cells, [-1.0]*3, [-1.0 + 0.5*g for g in grid])
    fp = adaflo_amd.FlowParameters(velocity_degree=k)
    ts = adaflo_amd.TimeStepping(fp); ts.next()
    op = parallel.DistributedNavierStokesMatrix(fp, part, device=0)
    op.initialize(ts, True)
    rng = np.random.default_rng(5)
    nu = [k*g*c+1 for g, c in zip(grid, cells)]; npn = [g*c+1 for g, c in zip(grid, cells)]
    gu = rng.uniform(-1, 1, (nu[2], nu[1], nu[0], 3)); gp = rng.uniform(-1, 1, (npn[2], npn[1], npn[0], 1))
    lu, lp = view(gu, part, k), view(gp, part, k-1)
    halo = op.halo
    pu = torch.from_numpy((lu / mult(part, k)).reshape(-1).copy()).to(dev)
    pp = torch.from_numpy((lp / mult(part, k-1)).reshape(-1).copy()).to(dev)
    halo.compress_add([pu, pp]); torch.cuda.synchronize()
    e_add = (rel_l2(pu.cpu().numpy(), lu.reshape(-1)), rel_l2(pp.cpu().numpy(), lp.reshape(-1)))
    ou, opm = halo.owned_mask(0).numpy(), halo.owned_mask(1).numpy()
    su = torch.from_numpy(np.where(ou > 0, lu.reshape(-1), 1e30)).to(dev)
    sp = torch.from_numpy(np.where(opm > 0, lp.reshape(-1), 1e30)).to(dev)
    halo.update_ghost_values([su, sp]); torch.cuda.synchronize()
    e_gh = (rel_l2(su.cpu().numpy(), lu.reshape(-1)), rel_l2(sp.cpu().numpy(), lp.reshape(-1)))
    # projection weights vs global
    print(rank, "add", e_add, "ghost", e_gh, "inv", float(op._inv.cpu()), flush=True)
    # distributed vmult with / without projection against the oracle on the global mesh
    from oracle import oracle as orc
    gmesh = orc.Mesh.make([g*c for g, c in zip(grid, cells)], [-1.0]*3, [-1.0 + 0.5*g for g in grid])
    glin = np.random.default_rng(11).uniform(-1, 1, (gmesh.ncell[2], gmesh.ncell[1], gmesh.ncell[0], 27*12))
    cs = tuple(slice(part.coords[d]*cells[d], (part.coords[d]+1)*cells[d]) for d in (2, 1, 0))
    op.local.set_linearization(np.ascontiguousarray(glin[cs]).reshape(-1))
    prm = orc.NSParams.make(weight=ts.weight(), weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old())
    gcon = orc.boundary_mask(gmesh, k, 3)
    gw = orc.ns_pressure_mass_weight(gmesh, k)
    V = adaflo_amd.DeviceVector.from_torch; ctx = op.local._ctx
    for fix in (False, True):
        op.pressure_average_fix = fix
        ref_u, ref_p = orc.ns_vmult(gmesh, k, prm, gu.reshape(-1).copy(), gp.reshape(-1).copy(), gcon, None, lin=glin.reshape(-1),
                                    weights=gw if fix else None, modes=np.ones_like(gw) if fix else None)
        su = torch.from_numpy(lu.reshape(-1).copy()).to(dev); sp = torch.from_numpy(lp.reshape(-1).copy()).to(dev)
        du, dp = torch.zeros_like(su), torch.zeros_like(sp)
        op.vmult(adaflo_amd.BlockVector([V(ctx, du), V(ctx, dp)]), adaflo_amd.BlockVector([V(ctx, su), V(ctx, sp)]))
        torch.cuda.synchronize()
        ru = view(ref_u.reshape(nu[2], nu[1], nu[0], 3), part, k).reshape(-1)
        rp = view(ref_p.reshape(npn[2], npn[1], npn[0], 1), part, k-1).reshape(-1)
        print(rank, "fix", fix, rel_l2(du.cpu().numpy(), ru), rel_l2(dp.cpu().numpy(), rp), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = int(sys.argv[1]); grid = parallel.brick_grid(world)
    mp.spawn(worker, args=(world, port, grid, [8, 5, 6]), nprocs=world, join=True)
