"""
`ParallelTempering` — counterpart of `inference.mcmc.parallel.ParallelTempering`
(reference: inference/mcmc/parallel.py:69-384): a ladder of chains at increasing
temperature which periodically propose pairwise position swaps.

Same class surface (`take_steps`, `swap`, `advance`, `run_for`, `tight_pairs`,
`uniform_pairs`, `return_chains`, `shutdown`, swap counters) and the same swap
rule, U <= exp(-(b_i - b_j)(p_i / b_i - p_j / b_j)) with the swapped log-probability
re-tempered by the receiving chain (parallel.py:207-231, :62).

Mechanism differs: the reference spawns one OS process per chain and talks to it
over pipes (parallel.py:127-136); here the chains live in this process and advance
in lockstep so that each proposal round of the whole ladder is one batched
log-marginal-likelihood evaluation on the GPU.  Several ladders are advanced
together by `advance_ladders`; ladders are the unit sharded over GPUs (every swap
stays GPU-local, nothing but the final samples is gathered).
`swap_diagnostics` (matplotlib) is out of scope.
"""
import os
import sys
from collections import deque
from random import choice
from time import time
from warnings import warn

from numpy import arange, array, exp, identity, zeros
from numpy.random import default_rng

from inference_amd.mcmc.gibbs import advance_lockstep


def _common_batch_posterior(chains):
    """If every chain's posterior is `marginal_likelihood` of one device-backed GpRegressor,
    return its batched form."""
    owner = None
    for ch in chains:
        fn = getattr(ch, "posterior", None)
        obj = getattr(fn, "__self__", None)
        if obj is None or getattr(fn, "__name__", "") != "marginal_likelihood":
            return None
        if not hasattr(obj, "marginal_likelihood_batch") or (owner is not None and obj is not owner):
            return None
        owner = obj
    if owner is None:
        return None
    if hasattr(owner, "batch_independent_values"):
        owner.batch_independent_values(True)  # ragged retry rounds: a value must not depend on its batch
    return owner.marginal_likelihood_batch


class ParallelTempering:
    """
    :param chains: chain objects (e.g. `GibbsChain`) sorted by increasing temperature.
    :param batch_posterior: optional `thetas (B, P) -> (B,)` batched form of the chains'
        posterior; detected automatically for `GpRegressor.marginal_likelihood`.
    """

    def __init__(self, chains, batch_posterior=None):
        self.rng = default_rng()
        # pairing uses the stdlib generator (the module-level `random.choice`, parallel.py:172); a ladder that must
        # not share that global stream with other ladders (sharded runs) gets its own `random.Random(seed).choice`
        self.pair_choice = choice
        self.chains = list(chains)
        self.temperatures = [1.0 / chain.inv_temp for chain in self.chains]
        self.inv_temps = [chain.inv_temp for chain in self.chains]
        self.N_chains = len(self.chains)
        self.attempted_swaps = identity(self.N_chains)
        self.successful_swaps = zeros([self.N_chains, self.N_chains])
        self.batch_posterior = batch_posterior or _common_batch_posterior(self.chains)
        self.posterior_evaluations = 0
        if sorted(self.temperatures) != self.temperatures:
            warn(
                """
                The list of Markov-chain objects passed to ParallelTempering
                should be sorted in order of increasing chain temperature.
                """
            )

    def take_steps(self, n: int):
        """Advance all the chains `n` steps without performing any swaps."""
        if self.batch_posterior is not None:
            self.posterior_evaluations += advance_lockstep(self.chains, n, self.batch_posterior)
        else:
            for chain in self.chains:
                for _ in range(n):
                    chain.take_step()

    def uniform_pairs(self):
        proposed = arange(self.N_chains)
        self.rng.shuffle(proposed)
        return list(zip(proposed[::2], proposed[1::2]))

    def tight_pairs(self):
        """Random disjoint pairing in which almost every pair is one or two temperature levels apart
        (behaviour of parallel.py:162-188, random numbers consumed in the same order: one stdlib `choice` per
        picked pair over the candidates still free, then one shuffle of the chains left unpaired)."""
        n = self.N_chains
        # neighbours at distance 1 and 2 in ladder order; the last candidate (n - 2, n) would leave the ladder
        candidates = []
        for low in range(n - 1):
            candidates += [(low, low + 1), (low, low + 2)]
        candidates.pop()
        chosen, taken = [], set()
        while candidates:
            pick = self.pair_choice(candidates)
            chosen.append(pick)
            taken.update(pick)
            candidates = [c for c in candidates if c[0] not in taken and c[1] not in taken]
        if len(chosen) != n // 2:
            # the greedy pass can strand chains that are not neighbours: pair them at random
            free = [i for i in range(n) if i not in taken]
            self.rng.shuffle(free)
            chosen += [(min(a, b), max(a, b)) for a, b in zip(free[::2], free[1::2])]
        return chosen

    def draw_swap_plan(self):
        """The random part of one `swap()`: the pairs and one uniform per pair, consumed from the ladder's generators in
        swap()'s own order (pairing first, then one `rng.random()` per pair).  None of it depends on the chains' states,
        so a driver may draw it ahead and apply the pairs one by one (`apply_pair`) as their chains become ready."""
        proposed_swaps = self.tight_pairs()
        for pair in proposed_swaps:
            self.attempted_swaps[pair] += 1
        return proposed_swaps, [self.rng.random() for _ in proposed_swaps]

    def apply_pair(self, i: int, j: int, u: float):
        """Swap rule of parallel.py:207-231 for one pair, with the uniform `u` drawn for it."""
        ci, cj = self.chains[i], self.chains[j]
        dt = self.inv_temps[i] - self.inv_temps[j]
        pi = ci.probs[-1] / self.inv_temps[i]
        pj = cj.probs[-1] / self.inv_temps[j]
        dp = pi - pj
        if u <= exp(-dt * dp):
            pos_i, pos_j = ci.get_last(), cj.get_last()
            ci.replace_last(pos_j)
            ci.probs[-1] = pj * ci.inv_temp
            cj.replace_last(pos_i)
            cj.probs[-1] = pi * cj.inv_temp
            self.successful_swaps[i, j] += 1

    def swap(self):
        """Propose a position swap between randomly paired chains (parallel.py:190-231).  (The pairs are disjoint, so
        applying them one after another on the chains' current states is the reference's snapshot-then-apply.)"""
        pairs, uniforms = self.draw_swap_plan()
        for (i, j), u in zip(pairs, uniforms):
            self.apply_pair(i, j, u)

    def advance(self, n: int, swap_interval=10, display_progress=False):
        """Advance each chain by `n` steps with swap attempts every `swap_interval` steps."""
        total_cycles = n // swap_interval
        t_start = time()
        for j in range(total_cycles):
            self.take_steps(swap_interval)
            self.swap()
            if display_progress:
                pct = int(100 * (j + 1) / total_cycles)
                eta = int((time() - t_start) * (total_cycles / (j + 1) - 1))
                sys.stdout.write(f"\r  [ Running ParallelTempering - {pct}% complete   ETA: {eta} sec ]    ")
                sys.stdout.flush()
        if n % swap_interval != 0:
            self.take_steps(n % swap_interval)
        if display_progress:
            sys.stdout.write("\r  [ Running ParallelTempering - complete! ]                    \n")

    def run_for(self, minutes=0, hours=0, swap_interval=10):
        end_time = time() + (hours * 60.0 + minutes) * 60.0
        while time() < end_time:
            self.take_steps(swap_interval)
            self.swap()

    def return_chains(self):
        return self.chains

    def shutdown(self):
        """Nothing to stop: there are no worker processes."""


def advance_ladders(ladders, n: int, swap_interval=10, batch_posterior=None):
    """Advance several ParallelTempering ladders together: the chains of all ladders propose in lockstep (one batched
    device evaluation per proposal round), and every ladder performs its own swaps.  This is the per-GPU unit of
    config 5 (whole ladders per GPU, swaps GPU-local).

    Nobody waits for more than ONE other chain (round 4).  A swap point pairs the chains of a ladder at random
    (parallel.py:162-188) - and neither the pairing nor the uniform numbers of the accept tests depend on the chains'
    states, so the ladder draws them when its first chain arrives (`ParallelTempering.draw_swap_plan`: the same numbers,
    from the same generators, in the same order as `swap()` draws them) and every pair is swapped as soon as ITS two
    chains have finished the interval; both then start the next interval while the rest of the ladder is still in
    this one.  (Round 2 stopped every ladder at every swap point, round 3 every chain of a ladder until the ladder's
    slowest - the retry-until-accept loop of gibbs.py:635-648 makes the slowest of eight chains need ~1.3x the average
    number of proposals - had arrived: the batches thinned out towards every swap point.)  A chain draws from its own
    generators, a ladder's swaps from the ladder's, and a pair's swap touches its two chains only: every trajectory and
    every swap decision is what `ParallelTempering.advance` gives for the ladder run alone; results do not depend on how
    ladders are grouped or sharded.  (Ladders that still pair their chains with the module-level `random.choice` of
    parallel.py:172 share ONE stream; for those every swap point stays a common one and the swaps are made in ladder
    order, as before.)
    Returns the number of posterior evaluations made."""
    bp = batch_posterior or ladders[0].batch_posterior
    if bp is None:
        raise ValueError("advance_ladders needs a batched posterior")
    if n <= 0 or not ladders:
        return 0
    chains = [c for lad in ladders for c in lad.chains]
    owner = [k for k, lad in enumerate(ladders) for _ in lad.chains]
    P = chains[0].n_parameters
    first_of = []
    k0 = 0
    for lad in ladders:
        first_of.append(k0)
        k0 += len(lad.chains)
    n_swaps = n // swap_interval                    # full intervals: each ends with a swap (as in `advance`)
    n_intervals = n_swaps + (1 if n % swap_interval else 0)

    def length(t):  # steps of interval t (a trailing partial interval ends without a swap)
        return swap_interval if t < n_swaps else n - n_swaps * swap_interval

    tc = [0] * len(chains)     # interval the chain is in
    step = [0] * len(chains)   # steps taken inside it
    par = [0] * len(chains)
    p_old = [c.probs[-1] for c in chains]
    p_acc = list(p_old)
    prop = [c.get_last() for c in chains]
    common = len(ladders) > 1 and any(lad.pair_choice is choice for lad in ladders)  # a shared random stream

    def _restart(c):
        """Chain c enters its next interval; False if it has none left."""
        tc[c] += 1
        if tc[c] >= n_intervals:
            return False
        step[c] = 0
        par[c] = 0
        p_old[c] = chains[c].probs[-1]
        p_acc[c] = p_old[c]
        prop[c] = chains[c].get_last()
        return True

    # swap plans, one per ladder and swap point, drawn in order when the first chain reaches the swap point
    plans = [[] for _ in ladders]    # plans[k][t] = {local chain: (partner, pair, uniform)}
    arrived = [dict() for _ in ladders]  # arrived[k][t] = set of local chains waiting for their partner

    def _plan(k, t):
        while len(plans[k]) <= t:
            pairs, us = ladders[k].draw_swap_plan()
            m = {}
            for (i, j), u in zip(pairs, us):
                m[int(i)] = (int(j), (int(i), int(j)), u)
                m[int(j)] = (int(i), (int(i), int(j)), u)
            plans[k].append(m)
        return plans[k][t]

    def _interval_done(c, still):
        """Chain c has finished its interval: its pair swaps once both are there; whoever can goes on."""
        k, t = owner[c], tc[c]
        if t >= n_swaps:  # trailing partial interval: no swap, and nothing after it
            _restart(c)
            return
        i = c - first_of[k]
        entry = _plan(k, t).get(i)
        if entry is None:  # odd ladder: this chain sits the swap out
            if _restart(c):
                still.append(c)
            return
        j, pair, u = entry
        waiting = arrived[k].setdefault(t, set())
        if j in waiting:
            waiting.discard(j)
            ladders[k].apply_pair(pair[0], pair[1], u)
            for cc in (c, first_of[k] + j):
                if _restart(cc):
                    still.append(cc)
        else:
            waiting.add(i)

    def _settle(active, vals):
        """The accept / reject bookkeeping of one round: returns the chains that propose again (chains whose pair
        is complete included), and - for the common-stream mode - the chains that finished their interval."""
        still, ended = [], []
        for c, v in zip(active, vals):
            chain = chains[c]
            p_new = float(v) * chain.inv_temp
            if chain._mh_test(chain.params[par[c]], p_new, p_old[c]):
                p_old[c] = p_new
                p_acc[c] = p_new
                par[c] += 1
                if par[c] == P:
                    chain._commit(prop[c], p_acc[c])
                    step[c] += 1
                    par[c] = 0
                    if step[c] < length(tc[c]):
                        p_old[c] = chain.probs[-1]
                        p_acc[c] = p_old[c]
                        prop[c] = chain.get_last()
            if step[c] < length(tc[c]):
                still.append(c)
            elif common:
                ended.append(c)
            else:
                _interval_done(c, still)
        return still, ended

    evals = 0
    active = list(range(len(chains)))
    if common:
        # one stream for every ladder's pairing: all chains meet at every swap point, the ladders swap in ladder order
        held = []
        while active or held:
            if not active:
                t = tc[held[0]]
                if t < n_swaps:
                    for lad in ladders:
                        lad.swap()
                active = sorted(c for c in held if _restart(c))
                held = []
                continue
            for c in active:
                prop[c][par[c]] = chains[c].params[par[c]].proposal()
            vals = bp(array([prop[c] for c in active]))
            evals += len(active)
            active, ended = _settle(active, vals)
            held += ended
            active.sort()
        return evals
    # Two (or more) groups of whole ladders, evaluated in turn through the model's two asynchronous slots: while the
    # device works on one group's proposals the host settles the other's - the bookkeeping above is ~10 us per chain
    # and round, a sixth of the device time of a round of 64 chains.  Ladders never interact (a chain's partner is in
    # its own ladder, hence in its own group), generators are per chain and per ladder, and a value does not depend on
    # its batch: the trajectories are those of the one-batch loop below.
    model = getattr(bp, "__self__", None)
    gmax = 0
    if (len(ladders) > 1 and getattr(bp, "__name__", "") == "marginal_likelihood_batch"
            and hasattr(model, "marginal_likelihood_batch_submit") and model.async_batches()):
        gmax = min(getattr(model.engine, "ASYNC_MAX", 256), model.engine.async_slot_capacity())
    if gmax >= max(len(lad.chains) for lad in ladders) and os.environ.get("GPMI_PT_ASYNC", "1") != "0":
        if hasattr(model, "batch_independent_values"):
            model.batch_independent_values(True)  # (a round of one chain must take the same path as a round of many)
        groups, cur = [], []
        per = max(len(lad.chains) for lad in ladders)
        want = max(2, -(-len(chains) // gmax))  # groups needed
        size = -(-len(ladders) // want)         # ladders per group
        while size * per > gmax:
            size -= 1
        for k in range(len(ladders)):
            cur.append(k)
            if len(cur) == size:
                groups.append(cur)
                cur = []
        if cur:
            groups.append(cur)
        act = [[c for k in g for c in range(first_of[k], first_of[k] + len(ladders[k].chains))] for g in groups]
        ready = deque(range(len(groups)))  # groups with work, not in flight
        flying = deque()                   # (slot, group), oldest first
        free = [0, 1]
        try:
            while ready or flying:
                while ready and free:
                    g = ready.popleft()
                    for c in act[g]:
                        prop[c][par[c]] = chains[c].params[par[c]].proposal()
                    slot = free.pop(0)
                    model.marginal_likelihood_batch_submit(array([prop[c] for c in act[g]]), slot)
                    flying.append((slot, g))
                slot, g = flying.popleft()
                vals = model.marginal_likelihood_batch_wait(slot)
                free.append(slot)
                evals += len(act[g])
                still, _ = _settle(act[g], vals)
                still.sort()
                act[g] = still
                if still:
                    ready.append(g)
        finally:
            # an exception (a KeyboardInterrupt in a long run) between submit and wait must not leave a slot pending:
            # every later batch call on this model would be refused
            while flying:
                slot, _ = flying.popleft()
                try:
                    model.marginal_likelihood_batch_wait(slot)
                except Exception:
                    pass
        return evals
    while active:
        for c in active:
            prop[c][par[c]] = chains[c].params[par[c]].proposal()
        vals = bp(array([prop[c] for c in active]))
        evals += len(active)
        active, _ = _settle(active, vals)
        active.sort()  # a fixed order of the batch rows (values do not depend on it; the order of host work does)
    return evals
