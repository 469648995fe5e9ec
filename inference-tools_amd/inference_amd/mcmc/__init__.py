from inference_amd.mcmc.gibbs import GibbsChain, advance_lockstep
from inference_amd.mcmc.parallel import ParallelTempering, advance_ladders

__all__ = ["GibbsChain", "ParallelTempering", "advance_lockstep", "advance_ladders"]
