"""Independent Markov chains, one per GPU, and the only exchange on the path.

The leapfrog path itself never communicates: rank r runs chain r on device r.  ONCE PER SAMPLE -- a fixed point
every rank reaches the same number of times, unlike the attempt loop, which each rank leaves at its own acceptance --
each rank contributes the records {epsilon, accepted, neps} of the attempts of the sample it just finished to an
all-gather.  Every rank then appends the OTHER ranks' records to its step-size / acceptance ring, the two 100-entry
tables the reference adapts epsilon from (``acc_flag_N_a`` / ``epsilon_N_a``, ``barlib/include/struct_main.h:172-173``,
written by ``update_epsilon_acc_rate_tables``, ``barlib/src/hmc/leapfrog/time_step.cpp:187-203``), so the tables fill
``world_size`` times faster during burn-in.  With ``pool=False`` nothing is exchanged and each chain behaves exactly
like the single-chain reference.

Packing, queueing (a sample with more than ``BCHMC_EPS_BATCH`` attempts sends the rest later) and validation live
behind the C ABI (``bchmc_eps_exchange``, ``include/bchmc.h``); the transport is either RCCL inside the library
(``transport="rccl"``: ``ncclAllGather`` of 520 bytes per rank on a side stream, unique id broadcast through
``torch.distributed``) or ``torch.distributed.all_gather_into_tensor`` handed to the library as its custom
transport (``"torch"``: nccl = RCCL on GPUs, gloo on CPU in the tests).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import engine as _engine

RECORD_BYTES = 16  # bchmc_eps_record in include/bchmc.h: double epsilon; int32 accepted; int32 neps


class EpsRing:
    """The reference's acceptance / epsilon tables (time_step.cpp:187-203): a ring of N_a_eps_update entries
    indexed by (records - 1) % N_a_eps_update, records = this chain's attempts plus pooled ones (for a single chain:
    the reference's count_attempts)."""

    def __init__(self, n_a_eps_update=100):
        self.n = int(n_a_eps_update)
        self.acc_flag = np.zeros(self.n, dtype=bool)
        self.epsilon = np.zeros(self.n)
        self.count_attempts = 0   # records written so far
        self._checked = 0         # count_attempts at the previous crossed() call

    def record(self, accepted, epsilon):
        """One finished attempt: ``count_attempts`` was already incremented by Hamiltonian_EoM (HMC.cc:368)."""
        self.count_attempts += 1
        ix = (self.count_attempts - 1) % self.n
        self.acc_flag[ix] = bool(accepted)
        self.epsilon[ix] = float(epsilon)

    def acceptance_rate(self):
        """bool_mean(acc_flag_N_a), time_step.cpp:24-28."""
        return float(np.count_nonzero(self.acc_flag)) / self.n

    def crossed(self, every):
        """``count_attempts % every == 0 and count_attempts > 0`` (time_step.cpp:115-116, 168-169) as a crossing test:
        has the record count passed a multiple of ``every`` since the previous call?  Identical to the reference's
        equality for a single chain (one record per call); with pooled records the count advances by several between
        calls and equality would fire only at lcm(world, every)."""
        before, self._checked = self._checked, self.count_attempts
        return every > 0 and self.count_attempts // every != before // every

    def due_for_update(self):
        """time_step.cpp:115-116 (stateless form, single chain only)."""
        return self.count_attempts > 0 and self.count_attempts % self.n == 0


class ChainGroup:
    """Rank bookkeeping for independent chains plus the per-sample step-size-statistics exchange."""

    def __init__(self, pool=True, device=None, transport="torch"):
        self.pool = pool
        self.enabled = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.enabled else 0
        self.world = dist.get_world_size() if self.enabled else 1
        self.device = device if device is not None else torch.device("cpu")
        self.transport = transport
        self.comm = None
        if self.enabled and self.pool and self.world > 1:
            if transport == "rccl":
                # rank 0 draws the ncclUniqueId, torch.distributed carries the 128 bytes to the others
                uid = [_engine.Comm.unique_id() if self.rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                dev = self.device.index if self.device.type == "cuda" else 0
                self.comm = _engine.Comm(self.rank, self.world, device=dev, unique_id=uid[0])
            elif transport == "torch":
                # persistent buffers: the collective is latency-bound, never allocate in the loop
                self._send = torch.zeros(_engine.PACKET_BYTES, dtype=torch.uint8, device=self.device)
                self._recv = torch.zeros(_engine.PACKET_BYTES * self.world, dtype=torch.uint8, device=self.device)
                self.comm = _engine.Comm(self.rank, self.world, allgather=self._allgather)
            else:
                raise ValueError("transport must be 'torch' or 'rccl'")

    def _allgather(self, payload):
        self._send.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
        dist.all_gather_into_tensor(self._recv, self._send)
        return self._recv.cpu().numpy().tobytes()

    def chain_seed(self, seed):
        """Different seeds per chain: seed + rank."""
        return int(seed) + self.rank

    def exchange(self, records):
        """ONE all-gather (call it once per sample on every rank): ``records`` = [(epsilon, accepted, neps), ...] of
        this rank's attempts since the last call.  Returns [(rank, epsilon, accepted, neps), ...] of all ranks' records
        that travelled in this exchange (own ones included; without pooling: just the own ones)."""
        recs = [(float(e), bool(a), int(n)) for e, a, n in records]
        if self.comm is None:
            return [(self.rank,) + r for r in recs]
        return self.comm.exchange(recs)

    def pool_into(self, ring, records):
        """Exchange this sample's records and append the OTHER chains' ones to ``ring`` (the own attempts were
        recorded as they happened, like the reference does).  Returns what the exchange delivered."""
        got = self.exchange(records)
        for rk, eps, acc, _ in got:
            if rk != self.rank:
                ring.record(acc, eps)
        return got

    def broadcast_eps_fac(self, eps_fac):
        """Optional: share rank 0's eps_fac after an adjustment so all chains use one step size."""
        if not (self.enabled and self.pool) or self.world == 1:
            return float(eps_fac)
        t = torch.tensor([float(eps_fac)], dtype=torch.float64, device=self.device)
        dist.broadcast(t, src=0)
        return float(t.item())

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
