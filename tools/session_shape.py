"""Prints the slot geometry a bench-like Session really gets (GPU box): group size, slots of the pooled sampler."""
import ctypes as C
import os
import sys
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from salient_plusplus_amd import _native as nat
from salient_plusplus_amd import fast_sampler as fs
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
from salient_plusplus_amd.synthetic import make_workload
dev = torch.device("cuda", 0)
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "S-products", seed=1234, device=dev)
cfg = FastSamplerConfig(x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
                        batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False,
                        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
                        exact_num_batches=wl.train_idx.numel() // wl.batch_size, count_remote_frequency=False, use_cache=False)
slots = int(os.environ.get("SPP_MAX_SLOTS", "32"))
it = iter(FastSampler(4, slots, cfg))
s = it.session
L = nat.load()
sc = nat.SamplerCfg()
L.spp_sampler_get_cfg(L.spp_session_sampler(s._h), C.byref(sc))
print("requested slots", slots, "group size", L.spp_session_group_size(s._h), "sampler slots", sc.num_slots)
