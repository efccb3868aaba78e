"""fmcmc_amd: MI355X-native many-chain Metropolis-Hastings engine, a drop-in for the hot path of
USCbiostats/fmcmc (MCMC(), kernel_*(), convergence_gelman()).  See DESIGN.md / INTEGRATION.md."""
from .kernels import (kernel_normal, kernel_normal_reflective, kernel_adapt, kernel_am, kernel_ram, kernel_unif,
                      kernel_unif_reflective, kernel_nmirror, kernel_umirror, fmcmc_kernel, plan_update_sequence, check_dimensions, process_bounds,
                      eta_power, qfun_t, qfun_normal)
from .models import gaussian_linreg, logistic, iid_normal, LogPosterior
from .mcmc import (MCMC, MCMC_without_conv_checker, MCMC_with_conv_checker, Mcmc, McmcList, check_initial,
                   append_chains, get_logpost, get_draws, get_elapsed, shard_bounds, DeviceChains, get_, get_initial,
                   get_fun, get_nsteps, get_seed, get_nchains, get_burnin, get_thin, get_kernel, get_multicore,
                   get_conv_checker, get_cl, get_progress, get_chain_id, MCMC_OUTPUT)
from .convergence import (convergence_gelman, convergence_geweke, convergence_heildel, convergence_auto, geweke_diag,
                          heidel_diag, spectrum0_ar)
from .recursive import cov_recursive, mean_recursive, reflect_on_boundaries

__all__ = ["MCMC", "MCMC_without_conv_checker", "MCMC_with_conv_checker", "kernel_normal",
           "kernel_normal_reflective", "kernel_adapt", "kernel_am", "kernel_ram", "kernel_unif",
           "kernel_unif_reflective", "kernel_nmirror", "kernel_umirror", "gaussian_linreg",
           "logistic", "iid_normal", "convergence_gelman", "convergence_geweke", "convergence_heildel", "convergence_auto",
           "geweke_diag", "heidel_diag", "spectrum0_ar", "Mcmc", "McmcList", "check_initial",
           "append_chains", "get_logpost", "get_draws", "get_elapsed", "shard_bounds", "cov_recursive", "mean_recursive",
           "reflect_on_boundaries", "plan_update_sequence", "get_", "get_initial", "get_fun", "get_nsteps", "get_seed",
           "get_nchains", "get_burnin", "get_thin", "get_kernel", "get_multicore", "get_conv_checker", "get_cl", "get_progress",
           "get_chain_id", "MCMC_OUTPUT", "eta_power", "qfun_t", "qfun_normal"]
