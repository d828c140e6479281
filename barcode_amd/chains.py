"""Independent Markov chains, one per GPU, and the only collective on the path.

The leapfrog path itself never communicates: rank r runs chain r on device r.  Once per trajectory each
rank contributes one 16-byte record {epsilon, accepted, neps} to an all-gather (RCCL over xGMI when the
process group backend is "nccl", gloo on CPU for tests).  Every rank then appends ALL records to its
step-size / acceptance ring, the two 100-entry tables the reference adapts epsilon from
(``acc_flag_N_a`` / ``epsilon_N_a``, ``barlib/include/struct_main.h:172-173``, written by
``update_epsilon_acc_rate_tables``, ``barlib/src/hmc/leapfrog/time_step.cpp:187-203``), so the tables fill
``world_size`` times faster during burn-in.  With ``pool=False`` the exchange is skipped and each chain
behaves exactly like the single-chain reference.
"""
import struct

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 16  # bchmc_eps_record in include/bchmc.h: double epsilon; int32 accepted; int32 neps


class EpsRing:
    """The reference's acceptance / epsilon tables (time_step.cpp:187-203): a ring of N_a_eps_update entries
    indexed by (count_attempts - 1) % N_a_eps_update."""

    def __init__(self, n_a_eps_update=100):
        self.n = int(n_a_eps_update)
        self.acc_flag = np.zeros(self.n, dtype=bool)
        self.epsilon = np.zeros(self.n)
        self.count_attempts = 0

    def record(self, accepted, epsilon):
        """One finished attempt: ``count_attempts`` was already incremented by Hamiltonian_EoM (HMC.cc:368)."""
        self.count_attempts += 1
        ix = (self.count_attempts - 1) % self.n
        self.acc_flag[ix] = bool(accepted)
        self.epsilon[ix] = float(epsilon)

    def acceptance_rate(self):
        """bool_mean(acc_flag_N_a), time_step.cpp:24-28."""
        return float(np.count_nonzero(self.acc_flag)) / self.n

    def due_for_update(self):
        """time_step.cpp:115-116."""
        return self.count_attempts > 0 and self.count_attempts % self.n == 0


def pack_record(epsilon, accepted, neps):
    return struct.pack("<dii", float(epsilon), int(bool(accepted)), int(neps))


def unpack_record(buf):
    eps, acc, neps = struct.unpack("<dii", bytes(buf))
    return eps, bool(acc), neps


class ChainGroup:
    """Rank bookkeeping for independent chains plus the epsilon-statistics exchange."""

    def __init__(self, pool=True, device=None):
        self.pool = pool
        self.enabled = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.enabled else 0
        self.world = dist.get_world_size() if self.enabled else 1
        self.device = device if device is not None else torch.device("cpu")
        # persistent buffers: the collective is latency-bound, never allocate in the loop
        self._send = torch.zeros(RECORD_BYTES, dtype=torch.uint8, device=self.device)
        self._recv = torch.zeros(RECORD_BYTES * self.world, dtype=torch.uint8, device=self.device)

    def chain_seed(self, seed):
        """Different seeds per chain: seed + rank."""
        return int(seed) + self.rank

    def exchange(self, epsilon, accepted, neps):
        """All-gather one record per rank; returns the list of (epsilon, accepted, neps) in rank order."""
        mine = (float(epsilon), bool(accepted), int(neps))
        if not (self.enabled and self.pool) or self.world == 1:
            return [mine]
        payload = torch.frombuffer(bytearray(pack_record(*mine)), dtype=torch.uint8)
        self._send.copy_(payload)
        dist.all_gather_into_tensor(self._recv, self._send)
        raw = self._recv.cpu().numpy().tobytes()
        return [unpack_record(raw[r * RECORD_BYTES:(r + 1) * RECORD_BYTES]) for r in range(self.world)]

    def record_all(self, ring, epsilon, accepted, neps):
        """Exchange and append every chain's record to ``ring`` (own record only when pooling is off)."""
        recs = self.exchange(epsilon, accepted, neps)
        for eps, acc, _ in recs:
            ring.record(acc, eps)
        return recs

    def broadcast_eps_fac(self, eps_fac):
        """Optional: share rank 0's eps_fac after an adjustment so all chains use one step size."""
        if not (self.enabled and self.pool) or self.world == 1:
            return float(eps_fac)
        t = torch.tensor([float(eps_fac)], dtype=torch.float64, device=self.device)
        dist.broadcast(t, src=0)
        return float(t.item())
