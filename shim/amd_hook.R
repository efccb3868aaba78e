# amd_hook.R -- the R side of the drop-in: what fmcmc adds next to R/mcmc.R to run its hot path on an MI355X.
#
# Replaces the body of MCMC_without_conv_checker (R/mcmc.R:485-838) for ALL chains of a call when `fun` is one of the
# tagged closed-form families below and `kernel` is one of fmcmc's own kernels with default closures; everything else
# falls through to the interpreted path.  MCMC(), MCMC_with_conv_checker() (bulks, R/mcmc.R:841-1019), append_chains(),
# convergence_*(), get_logpost()/get_draws() and the kernel objects keep working unchanged, because this file
#   * writes the kernels' environments back exactly as the closures would have left them (R/kernel_normal.R:39-63,
#     R/kernel_adapt.R:87-170, R/kernel_ram.R:93-155, R/kernel.R:405-422 update_kernel),
#   * fills MCMC_OUTPUT$data.[[i]]$logpost / $draws with iteration names (R/mcmc.R:815-823),
#   * returns coda::mcmc / mcmc.list objects with the same mcpar (R/mcmc.R:829-836, :641).
# The C side is shim/fmcmc_amd_shim.c (`.Call`), the engine's C-ABI include/fmcmc_amd.h.
#
# Not run in this repository's image (no R); fmcmc_amd/mcmc.py + kernels.py are the executed mirror of this file and
# tests/test_gpu_api.py drives them with R's own random stream against the README's printed outputs.

# ---- tagged closed-form families: ordinary R functions (usable by the interpreted path) that carry their data ---------
gaussian_linreg <- function(X, y, intercept = TRUE, guard = TRUE) {          # README.md:128-139 (guard) / :356-361
  X <- if (is.null(X)) matrix(0, length(y), 0L) else as.matrix(X)
  storage.mode(X) <- "double"
  f <- function(p) {
    k  <- length(p)
    mu <- if (intercept) p[1L] + drop(X %*% p[seq_len(ncol(X)) + 1L]) else drop(X %*% p[seq_len(ncol(X))])
    ans <- sum(stats::dnorm(y - mu, sd = p[k], log = TRUE))
    if (guard && !is.finite(ans)) -Inf else ans
  }
  structure(f, class = c("fmcmc_amd_family", "function"),
            model = list(family = 1L, X = X, y = as.double(y), intercept = as.integer(intercept),
                         guard = as.integer(guard), prior_div = 0))
}
logistic <- function(X, y, intercept = TRUE, prior_div = 8) {                 # vignettes/workflow-with-fmcmc.Rmd:35-41
  X <- as.matrix(X); storage.mode(X) <- "double"
  f <- function(p) {
    eta  <- if (intercept) p[1L] + drop(X %*% p[-1L]) else drop(X %*% p)
    logp <- ifelse(eta < 0, eta - log1p(exp(eta)), -log1p(exp(-eta)))
    logq <- ifelse(eta < 0, -log1p(exp(eta)), -eta - log1p(exp(-eta)))
    sum(logp[y == 1]) + sum(logq[y == 0]) - (if (prior_div != 0) sum(p^2) / prior_div else 0)
  }
  structure(f, class = c("fmcmc_amd_family", "function"),
            model = list(family = 2L, X = X, y = as.double(y), intercept = as.integer(intercept), guard = 0L,
                         prior_div = as.double(prior_div)))
}
iid_normal <- function(D) {                                                   # R/mcmc.R:141-144
  f <- function(p) sum(log(stats::dnorm(D, p[1L], p[2L])))
  structure(f, class = c("fmcmc_amd_family", "function"),
            model = list(family = 3L, X = NULL, y = as.double(D), intercept = 1L, guard = 0L, prior_div = 0))
}

amd_available <- function() {
  info <- tryCatch(.Call("C_fmcmc_amd_info", PACKAGE = "fmcmc"), error = function(e) c(0L, 0L))
  info[2L] > 0L
}

# ---- which of fmcmc's kernels is this environment? ------------------------------------------------------------------
# The closures carry no type tag; the set of variables they keep does (R/kernel_*.R).  A kernel whose `eta` / `qfun`
# were replaced by arbitrary closures (R/kernel_ram.R:67-68) is not supported: they cannot run inside a GPU kernel.  The
# families those arguments are used for are offered as TAGGED closures: ordinary R functions (the interpreted path runs
# them unchanged) whose attributes tell the engine which built-in to use (fmcmc_kernel.ram_qfun / ram_df / ram_eta_exp).
amd_eta_power   <- function(exponent = 2 / 3) {
  stopifnot(is.finite(exponent), exponent > 0)
  structure(function(i, k) min(c(1.0, i^(-exponent) * k)), class = c("fmcmc_amd_eta", "function"), exponent = exponent)
}
amd_qfun_t      <- function(df) {
  stopifnot(is.finite(df), df > 0)
  structure(function(k) stats::rt(k, df), class = c("fmcmc_amd_qfun", "function"), ram_qfun = 2L, df = df)
}
amd_qfun_normal <- function()
  structure(function(k) stats::rnorm(k), class = c("fmcmc_amd_qfun", "function"), ram_qfun = 1L, df = 0)
amd_ram_families <- function(k) {          # list(ram_qfun, ram_df, ram_eta_exp), or NULL when a closure is not one of ours
  dflt <- formals(kernel_ram)
  q <- if (inherits(k$qfun, "fmcmc_amd_qfun")) list(attr(k$qfun, "ram_qfun"), attr(k$qfun, "df"))
       else if (isTRUE(all.equal(k$qfun, eval(dflt$qfun), check.environment = FALSE))) list(0L, 0) else return(NULL)
  e <- if (inherits(k$eta, "fmcmc_amd_eta")) attr(k$eta, "exponent")
       else if (isTRUE(all.equal(k$eta, eval(dflt$eta), check.environment = FALSE))) 0 else return(NULL)
  list(ram_qfun = as.integer(q[[1L]]), ram_df = as.double(q[[2L]]), ram_eta_exp = as.double(e))
}
amd_kernel_kind <- function(k) {
  v <- ls(k, all.names = TRUE)
  if (all(c("eta", "qfun", "arate") %in% v)) {
    if (is.null(amd_ram_families(k))) return(NA_integer_)
    return(4L)
  }
  if (all(c("bw", "Sd", "Mean_t_prev") %in% v)) return(3L)
  if ("nadapt" %in% v) return(if ("sqrt3" %in% v || isTRUE(k$is_uniform)) 8L else 7L)
  if (all(c("min.", "max.") %in% v)) return(if ("lb" %in% v) 6L else 5L)
  if (all(c("mu", "scale", "scheme") %in% v)) return(if ("lb" %in% v) 2L else 1L)
  NA_integer_
}
fmcmc_amd_supported <- function(kernel) {
  ks <- if (is_kernel_list(kernel)) lapply(seq_along(kernel), function(i) kernel[[i]]) else list(kernel)
  all(!is.na(vapply(ks, amd_kernel_kind, 1L)))
}

# ---- kernel_spec(): the first-call initialisation of the closures, done once for the engine --------------------------
# (check_dimensions / process_bounds / which. / plan_update_sequence: R/kernel_normal.R:39-63, R/kernel_adapt.R:87-115,
#  R/kernel_ram.R:93-121, R/kernel_unif.R:46-70, R/kernel_mirror.R:66-100.)  `k1` is the FIRST kernel of the list: all
# chains share the constructor's arguments, only their state differs.
kernel_spec <- function(k1, k, nsteps) {
  kind  <- amd_kernel_kind(k1)
  rec   <- function(x, name) { if (length(x) > 1L && length(x) != k) stop("Incorrect length of -", name, "-.", call. = FALSE)
                               if (length(x) == 1L && k > 1L) rep(x, k) else x }
  fixed <- as.logical(rec(k1$fixed, "fixed"))
  which. <- which(!fixed)
  if (!length(which.))
    stop("The number of parameters to update, i.e. not fixed, cannot be zero. Check the value -fixed- in the kernel initialization.",
         call. = FALSE)
  full <- function(x, name, default) {                 # a k-vector even when the closure already restricted it to which.
    x <- if (is.null(x)) default else x
    if (length(x) == length(which.) && length(which.) != k) { y <- rep(default[1L], k); y[which.] <- x; y } else rec(x, name)
  }
  big <- .Machine$double.xmax
  unif <- kind %in% c(5L, 6L)
  mu    <- if (unif) rec(k1$min., "min.") else if (kind == 4L) rep(0, k) else full(k1$mu, "mu", 0)
  scale <- if (unif) rec(k1$max., "max.") - rec(k1$min., "min.") else if (kind %in% c(3L, 4L)) rep(1, k) else rec(k1$scale, "scale")
  lb <- if (kind %in% c(1L, 5L)) rep(-big, k) else process_bounds(rec(k1$lb, "lb"), is_lower = TRUE)
  ub <- if (kind %in% c(1L, 5L)) rep(big, k) else process_bounds(rec(k1$ub, "ub"), is_lower = FALSE)
  if (!(kind %in% c(1L, 5L)) && any(ub <= lb)) stop("-ub- cannot be <= than -lb-.", call. = FALSE)
  if (unif && any(scale <= 0)) stop("-max.- cannot be <= than -min.-.", call. = FALSE)
  scheme <- 0L; scheme_seq <- NULL
  if (kind %in% c(1L, 2L, 5L, 6L, 7L, 8L)) {
    sc <- k1$scheme
    if (length(sc) > 1L && is.numeric(sc)) {
      plan_update_sequence(k, 1L, fixed, sc)           # the reference's own checks and messages (R/kernel.R:72-90)
      scheme <- 3L; scheme_seq <- as.integer(sc) - 1L
    } else scheme <- switch(as.character(sc), joint = 0L, ordered = 1L, random = 2L,
                            stop("-scheme- update must be either an integer sequence, 'joint', 'ordered', or 'random'.", call. = FALSE))
  }
  kf <- length(which.)
  constr <- if (kind == 4L && !is.null(k1$constr)) as.double(t(k1$constr[which., , drop = FALSE][, which., drop = FALSE])) else NULL
  ram <- if (kind == 4L) amd_ram_families(k1) else list(ram_qfun = 0L, ram_df = 0, ram_eta_exp = 0)
  list(kind = kind, k = as.integer(k), ram_qfun = ram$ram_qfun, ram_df = ram$ram_df, ram_eta_exp = ram$ram_eta_exp,
       mu = as.double(mu), scale = as.double(scale), lb = as.double(lb), ub = as.double(ub),
       fixed = fixed, scheme = scheme, scheme_seq = scheme_seq,
       freq = as.integer(if (is.null(k1$freq)) 1L else k1$freq), warmup = as.integer(if (is.null(k1$warmup)) 0L else k1$warmup),
       bw = as.integer(if (is.null(k1$bw)) 0L else k1$bw), until = as.double(if (is.null(k1$until)) Inf else k1$until),
       eps = as.double(if (is.null(k1$eps)) 1e-4 else k1$eps), arate = as.double(if (is.null(k1$arate)) 0.234 else k1$arate),
       Sd = as.double(if (is.null(k1$Sd)) (if (kind == 3L) 5.76 / kf else 0) else k1$Sd),
       nadapt = as.integer(if (is.null(k1$nadapt)) 4L else k1$nadapt), constr = constr,
       which. = which., kf = kf)
}

# ---- kernel_state(): the persistent part of every chain's kernel environment -> fmcmc_state ---------------------------
kernel_state <- function(kernels, spec, initial) {
  C <- length(kernels); kf <- spec$kf; k <- spec$k
  used <- !vapply(kernels, function(e) is.null(e$amd_steps_done), TRUE)          # has the engine run this kernel before?
  st <- list(theta0 = as.double(t(initial)), fresh = as.integer(!all(used)))
  if (all(used)) {
    st$abs_iter <- as.double(vapply(kernels, function(e) if (is.null(e$abs_iter)) 0 else e$abs_iter, 0))
    if (spec$kind %in% c(3L, 4L)) {
      st$Sigma <- as.double(vapply(kernels, function(e) t(e$Sigma), matrix(0, kf, kf)))   # ABI: row-major per chain
      st$nerrors <- as.integer(vapply(kernels, function(e) if (is.null(e$nerrors)) 0L else as.integer(e$nerrors), 0L))
    }
    if (spec$kind == 3L) {
      st$have_mean <- as.integer(vapply(kernels, function(e) !is.null(e$Mean_t_prev), TRUE))
      st$mean_prev <- as.double(vapply(kernels, function(e) if (is.null(e$Mean_t_prev)) rep(0, kf) else as.double(e$Mean_t_prev), rep(0, kf)))
    }
    if (spec$kind %in% c(7L, 8L)) {
      st$mirror_mu    <- as.double(vapply(kernels, function(e) as.double(e$mu), rep(0, k)))
      st$mirror_scale <- as.double(vapply(kernels, function(e) as.double(e$scale), rep(0, k)))
      # (the closure's obs_arate: NULL before the one-off adaptation, a scalar after it, a k-vector -- or numeric(0) -- once the
      #  warm-up's element-wise mean_recursive has touched it, R/kernel_mirror.R:108-118; the engine keeps k entries, NA = none)
      st$obs_arate    <- as.double(vapply(kernels, function(e) if (length(e$obs_arate) == 0L) rep(NA_real_, k) else rep_len(as.double(e$obs_arate), k), rep(0, k)))
    }
    if (spec$scheme == 2L)                                  # the plan is made ONCE per kernel object (R/kernel.R:106-113)
      st$scheme_cols <- as.integer(vapply(kernels, function(e) as.integer(max.col(e$update_sequence, "first") - 1L),
                                          integer(nrow(kernels[[1L]]$update_sequence))))
  } else if (spec$kind %in% c(3L, 4L) && !is.null(kernels[[1L]]$Sigma)) {
    # user-supplied Sigma (R/kernel_adapt.R:106-107, R/kernel_ram.R:112-113): the engine starts from it instead of eps * I
    st$fresh <- 0L
    st$abs_iter <- rep(0, C); st$nerrors <- rep(0L, C)
    st$Sigma <- as.double(vapply(kernels, function(e) t(e$Sigma), matrix(0, kf, kf)))
    st$have_mean <- rep(0L, C); st$mean_prev <- rep(0, kf * C)
  }
  st
}

# ---- kernel_write_back(): leave every environment as its closure would have (R/kernel.R:405-422) --------------------
kernel_write_back <- function(kernels, spec, res, nsteps) {
  s <- res$state; kf <- spec$kf; k <- spec$k; w <- spec$which.
  for (c in seq_along(kernels)) {
    e <- kernels[[c]]
    e$fixed <- spec$fixed
    if (spec$kind %in% c(1L, 2L, 5L, 6L, 7L, 8L)) {
      if (spec$kind %in% c(5L, 6L)) { e$min. <- spec$mu; e$max. <- spec$mu + spec$scale } else if (spec$kind <= 2L) { e$mu <- spec$mu; e$scale <- spec$scale }
      if (!(spec$kind %in% c(1L, 5L))) { e$lb <- spec$lb; e$ub <- spec$ub }
      us <- if (is.null(e$update_sequence)) {
        if (spec$scheme == 2L) {                              # the engine drew the plan: column indices -> logical matrix
          m <- matrix(FALSE, nsteps, k); m[cbind(seq_len(nsteps), s$scheme_cols[, c] + 1L)] <- TRUE; m
        } else plan_update_sequence(k, nsteps, spec$fixed, e$scheme)
      } else e$update_sequence
      e$update_sequence <- us
      e$k <- sum(us[1L, ])                                     # k <<- sum(update_sequence[1, ]) (R/kernel_normal.R:61)
      if (spec$kind %in% c(7L, 8L)) {
        e$mu <- s$mirror_mu[, c]; e$scale <- s$mirror_scale[, c]; e$obs_arate <- if (all(is.na(s$obs_arate[, c]))) NULL else s$obs_arate[, c]; e$abs_iter <- s$abs_iter[c]
      }
    } else {
      e$k <- kf; e$which. <- w; e$lb <- spec$lb; e$ub <- spec$ub
      e$Ik <- if (spec$kind == 3L) diag(kf) * spec$eps else diag(kf)              # R/kernel_adapt.R:103, R/kernel_ram.R:107
      e$Sigma <- t(matrix(s$Sigma[, , c], kf, kf))              # the ABI is row-major
      e$abs_iter <- s$abs_iter[c]
      e$nerrors <- s$nerrors[c]
      if (spec$kind == 3L) {
        e$mu <- spec$mu[w]; e$Sd <- spec$Sd
        e$Mean_t_prev <- if (s$have_mean[c]) matrix(s$mean_prev[, c], nrow = 1L) else NULL   # 1 x k matrix, R/kernel_adapt.R:160-161
      }
    }
    e$amd_steps_done <- (if (is.null(e$amd_steps_done)) 0 else e$amd_steps_done) + nsteps    # Philox counter of the next call
  }
  invisible(NULL)
}

# ---- the hook: first lines of MCMC_without_conv_checker (R/mcmc.R:485) -------------------------------------------------
#   if (inherits(fun, "fmcmc_amd_family") && amd_available() && fmcmc_amd_supported(kernel))
#     return(amd_MCMC_without_conv_checker(initial, fun, nsteps, nchains, burnin, thin, kernel, chain_id))
amd_MCMC_without_conv_checker <- function(initial, fun, nsteps, nchains = 1L, burnin = 0L, thin = 1L,
                                          kernel = kernel_normal(), chain_id = 1L,
                                          seed = getOption("fmcmc.amd.seed", NULL), fed = NULL) {
  initial <- check_initial(initial, nchains)                       # R/checks.R:22-58 (names, recycling, warnings)
  k <- ncol(initial)
  if (nchains > 1L && !is_kernel_list(kernel)) rep_kernel(kernel, nchains = nchains)      # R/mcmc.R:526-527
  else if (nchains == 1L && is_kernel_list(kernel))
    stop("The passed kernel is for MCMC with more than one chain. Right now, -kernel- is of length ", length(kernel), call. = FALSE)
  kernels <- if (is_kernel_list(kernel)) lapply(seq_len(nchains), function(i) kernel[[i]]) else list(kernel)
  spec  <- kernel_spec(kernels[[1L]], k, nsteps)
  state <- kernel_state(kernels, spec, initial)
  # the engine's stream is keyed by (seed, chain, step): one draw from R's generator per MCMC() call keeps set.seed()
  # meaningful (R/mcmc.R:455-456) and makes bulks continue each other through step_base
  if (is.null(seed)) {
    seed <- kernels[[1L]]$amd_seed
    if (is.null(seed)) seed <- floor(stats::runif(1L) * 2^52)
  }
  for (e in kernels) e$amd_seed <- seed
  done <- kernels[[1L]]$amd_steps_done
  run <- list(nchains = nchains, nsteps = nsteps, burnin = burnin, thin = thin, seed = seed, chain_base = chain_id - 1L,
              step_base = if (is.null(done)) 0 else done, rng_mode = 0L, device = getOption("fmcmc.amd.device", 0L))
  if (!is.null(fed)) {                                             # bit-level replay of an interpreted run: R's own draws
    kz <- if (spec$kind %in% c(3L, 4L) || spec$scheme == 0L) spec$kf else 1L
    f <- fed(nchains, nsteps, kz, spec)                            # list(logu [nsteps x C], z [kz x nsteps x C])
    run$rng_mode <- 1L; run$fed_logu <- as.double(f$logu); run$fed_z <- as.double(f$z)
  }
  res <- .Call("C_fmcmc_amd_run", attr(fun, "model"), spec[setdiff(names(spec), c("which.", "kf"))], run, state, PACKAGE = "fmcmc")
  kernel_write_back(kernels, spec, res, nsteps)
  # results -> coda objects + MCMC_OUTPUT (R/mcmc.R:786-836)
  S <- dim(res$samples)[1L]
  iters <- burnin + thin * seq_len(S)
  cn <- colnames(initial)
  chains <- lapply(seq_len(nchains), function(c) {
    MCMC_OUTPUT$data.[[c]]$logpost <- structure(res$logpost[, c], names = iters)
    MCMC_OUTPUT$data.[[c]]$draws   <- matrix(res$draws[, , c], S, k, dimnames = list(iters, cn))
    coda::mcmc(matrix(res$samples[, , c], S, k, dimnames = list(iters, cn)), start = iters[1L], end = iters[S], thin = thin)
  })
  MCMC_OUTPUT$set_ptr(1L)
  if (nchains == 1L) chains[[1L]] else coda::as.mcmc.list(chains)
}
