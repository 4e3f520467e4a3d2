/* lmc_abi.h -- C ABI of the MI355X (gfx950) back end for the Langevin-MCMC chain loop.
 *
 * Drop-in boundary.  Two groups of entry points, all `extern "C"`, plain pointers and sizes:
 *
 * (1) The reference's own in-process plugin ABI: the shared object it dlopen()s as
 *     $DPT_LIBPATH/pathlibbidir_mala.so and resolves by name with dlsym
 *     (/root/reference/src/chad.cpp:884-895,1000-1016; names path.cpp:3404-3417; loop path.cpp:4028-4057):
 *         void evaluate_path_bidir_mala_<c>_<l>_static     (lens[2], primary[2L+1], scene[38], vertParams[V], logLum[1])
 *         void evaluate_path_bidir_mala_<c>_<l>_static_derv(lens[2], primary[2L+1], scene[38], vertParams[V], grad[2L])
 *     for 1<=c<=9, 0<=l<=8, 3<=c+l<=9 (L = c+l-1, V = 238+59(c+l-3)).  Caller type: PathFunc / PathFuncDerv
 *     (/root/reference/src/path.h:121-125); the caller passes a 6th NULL `hess` argument in MALA mode
 *     (mutation_mala.h:102-107), which these symbols ignore.  Each call launches the HIP kernel on one path;
 *     lmc_grad_batch is the entry point meant for throughput.
 *
 * (2) The batched / resident interface that replaces the per-chain loop of MLT() (mlt.cpp:20-214): scene
 *     load (ParseScene, parsescene.cpp:627-639), MLTInit (mlt.h:41-154), the chain loop (mlt.cpp:60-196),
 *     film read-back (mlt.cpp:203-207).
 *
 * Error convention: functions returning int give 0 on success, a negative value on failure with the message
 * available from lmc_last_error(); nothing throws across the boundary.  All functions fail (never fall back
 * to a CPU path) when no HIP device is usable.
 */
#ifndef LMC_ABI_H
#define LMC_ABI_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmc_ctx lmc_ctx;

/* Scene description: the reference's command line + <dpt> overrides (main.cpp:49-64, parsescene.cpp:535-590).
 * Integer fields <= 0 (seed_offset < 0) keep the value from the XML. */
typedef struct lmc_scene_desc {
    const char *scene_xml;  /* path of the Mitsuba-0.5-subset scene file, as given to `dpt` */
    int force_diffuse;      /* BASELINE.json config 2: every BSDF becomes `diffuse` */
    int max_depth;          /* <dpt maxdepth> override */
    int width, height;      /* film size override */
    int seed_offset;        /* --seedoffset (main.cpp:57-58) */
    int device;             /* HIP device ordinal */
    int use_gradient;       /* 1: evaluate d log f / d pss in-kernel (derivative library present);
                               0: behave like a missing pathlibbidir_mala.so (isotropic proposals, path.cpp:4042-4053) */
} lmc_scene_desc;

const char *lmc_last_error(void);

lmc_ctx *lmc_create(const lmc_scene_desc *desc);
void lmc_destroy(lmc_ctx *ctx);
/* number of HIP devices visible to this process (0 without a GPU): `device` of lmc_scene_desc must be below it */
int lmc_device_count(void);

/* Host-only (no GPU needed): what the front end parsed from a scene file (parsescene.cpp:535-639, loadserialized.cpp:153-325, parseobj.cpp:57-275,
 * image.cpp) as one JSON document: per mesh triangle / vertex counts, bounds, area, material, emitter; per material type and parameters
 * (textures: file, size, gamma, average); lights with their sampling weights and the light-pick CDF; camera; <dpt> options; output name.
 * Writes at most cap - 1 characters + NUL; returns the document's full length, -1 on error.  The cross-check surface of the parsers. */
long long lmc_scene_dump(const char *scene_xml, int force_diffuse, char *out, long long cap);
/* [width, height, numTriangles, maxDepth, numBvhNodes, bvhDepth, numLights, mala] */
int lmc_info(lmc_ctx *ctx, int *out8);
/* the 38-float scene block of the plugin ABI (scene.cpp:160-169) */
int lmc_scene_params(lmc_ctx *ctx, float *out38);
/* <dpt> float options by XML name: largestepprob, largestepscale, mala, uniformmixprob, mala-stepsize, mala-gn,
 * perturbstddev, mindepth, h2mc, uselightcoordinatesampling, largestepmultiplexed, samplecache (parsescene.cpp:538-585).  mala / h2mc /
 * samplecache / uselightcoordinatesampling select what the resident chain state and the launch plan are laid out for: the call itself succeeds
 * at any time, but once one of them differs from its value at lmc_chains_init, lmc_chains_step returns -1 until the chains are initialised again.
 * Back-end switches (no counterpart in the reference): "timing" (per-step HIP events for lmc_step_timing), "overlap" (side streams on / off),
 * "max-derivatives-depth" (main.cpp:59-60), "resort_every" / "resort_first" (period and first step of the full re-sort of the resident chains by
 * technique and screen position, device/relocate.hip; default 32 / 4, 0 = off; read at the next lmc_chains_init), "exp_resort" (measurement hook) */
int lmc_set_option(lmc_ctx *ctx, const char *name, double value);
/* <dpt> options as parsed: spp, numinitsamples, numchains, directspp, mindepth, maxdepth, largestepprob, largestepscale,
 * mala, h2mc, seedoffset (dptoptions.h:7-34); back-end state: bvh_quantised (the scene's hot launches walk the 64-byte quantised BVH nodes),
 * bvh_thick_flat_share (the figure that choice is made by) */
int lmc_get_option(lmc_ctx *ctx, const char *name, double *value);
/* film "filename" of the scene (outputName); the reference appends "_timeuse_<seconds>s.exr" (mlt.cpp:208) */
const char *lmc_output_name(lmc_ctx *ctx);
/* image files through the library's own codecs: EXR / PNG in (rgb == NULL: size only), RGB half ZIP EXR out (image.cpp:44-62) */
int lmc_image_read(const char *path, int *w, int *h, float *rgb);
int lmc_image_write_exr(const char *path, const float *rgb, int w, int h);

/* MLTInit (mlt.h:41-154) + chain set-up (mlt.cpp:60-90).  `init_threads` plays NumSystemCores(): init stream t is
 * seeded RNG(t + seedOffset) (mlt.h:67).  The chains [chain_begin, chain_end) of the n_chains_total chains live on
 * this device (multi-GPU: one range per rank; seeds are global chain ids, mlt.cpp:61-62).
 * samples_per_chain / chains_need_extra as computed at mlt.cpp:36-40. */
int lmc_chains_init(lmc_ctx *ctx, long long num_init_samples, int n_chains_total, int init_threads, int chain_begin, int chain_end,
                    long long samples_per_chain, long long chains_need_extra);
/* Multi-rank jobs: MLTInit is sharded by init stream (rank r runs streams [V r / R, V (r + 1) / R), V = init_threads) and the ranks
 * exchange what the seeding needs (contribution counts, scores, the checkpoints of the seeding samples), so that normalization and
 * every chain's init state equal the one-rank result bit for bit.  Ranks of an RCCL job (lmc_comm_init BEFORE lmc_chains_init) call
 * lmc_chains_init collectively; the n contexts of an in-process group (one per GPU, or several on one GPU for bring-up) are driven
 * by the two calls below (chains split into n contiguous equal ranges; their collectives are device copies). */
int lmc_group_chains_init(lmc_ctx **ctxs, int n, long long num_init_samples, int n_chains_total, int init_threads, long long samples_per_chain,
                          long long chains_need_extra);
int lmc_group_chains_step(lmc_ctx **ctxs, int n, int n_steps);
/* The film merge of such a group (the reference merges its per-thread films in-process, mlt.cpp:203-207): every member's device film (and
 * splat-weight sum) becomes the sum over the members, through peer copies -- lmc_film_allreduce without a communicator.  Once per stepped
 * film, like it.  *ms (may be NULL): wall time of the merge. */
int lmc_group_film_reduce(lmc_ctx **ctxs, int n, double *ms);
/* out4 = [distinct devices among the members, ordered pairs of distinct member devices, of which with direct peer access enabled
 * (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess, once, when the group is set up), host threads that drive the group's steps (one per
 * member; LMC_GROUP_THREADS=0: one for all)] */
int lmc_group_info(lmc_ctx **ctxs, int n, long long *out4);
/* host wall time (ms) this context's steps took to QUEUE (launches, event operations, the exchange's copies) since the last call, and the
 * number of steps: what a rank-step costs the host thread that drives it, to be read next to the step's GPU time */
int lmc_host_issue_timing(lmc_ctx *ctx, double *ms, long long *steps);
/* CPU test hooks (need no GPU): the host-side plan of the sharded MLTInit from the padded blocks the ranks all-gather.
 * lmc_shard_layout: out5 = [first stream, end stream, first sample, end sample, samples of the largest rank] of `rank`;
 * lmc_shard_counts_probe: rank_first[world + 1] = first contribution of every rank's block (and the total);
 * lmc_shard_plan_probe: per chain of the job the init sample that seeds it, the seeding contribution's technique (c * 16 + l) and
 * lsScore; owned_begin[world + 1]: rank r's samples seed the chains [owned_begin[r], owned_begin[r + 1]); normalization */
int lmc_shard_layout(int world, int rank, int init_threads, long long num_init_samples, long long *out5);
int lmc_shard_counts_probe(int world, int init_threads, long long num_init_samples, const unsigned char *padded_counts, long long max_local_samples,
                           unsigned long long *rank_first);
int lmc_shard_plan_probe(int world, int init_threads, long long num_init_samples, int n_chains_total, const unsigned char *padded_counts,
                         long long max_local_samples, const unsigned char *padded_cl, const float *padded_ls, long long max_local_contribs,
                         long long *seed_sample, unsigned char *seed_cl, float *seed_ls, int *owned_begin, float *normalization);
/* normalization = avgScore (mlt.cpp:46-47), number of init contributions */
int lmc_init_result(lmc_ctx *ctx, float *normalization, long long *num_contribs);
/* parity probe: every contribution MLTInit collected, in stream order: index of the init sample that produced it, technique as
 * c * 16 + l, lsScore; returns the total count (at most `cap` entries are written) */
long long lmc_init_contribs(lmc_ctx *ctx, long long cap, long long *sample, int *cl, float *ls);
/* advances every resident chain by n_steps mutations (the loop body mlt.cpp:91-170), lock step */
int lmc_chains_step(lmc_ctx *ctx, int n_steps);
/* blocks until all queued work is done */
int lmc_sync(lmc_ctx *ctx);
/* indirect film buffer, W*H*3 floats, un-normalised like indirectBuffer (mlt.cpp:54) */
int lmc_film_read(lmc_ctx *ctx, float *rgb);
int lmc_film_clear(lmc_ctx *ctx);
/* DirectLighting(scene, directBuffer) (direct.cpp:4-54): direct_spp samples per pixel of paths of length <= 2, tile RNG
 * streams seeded tileIndex + seedOffset; un-normalised W*H*3 buffer like directBuffer (mlt.cpp:33-34).  The image the
 * reference writes is direct / directSpp + indirect / spp (mlt.cpp:203-207). */
int lmc_direct_lighting(lmc_ctx *ctx, int direct_spp);
int lmc_direct_read(lmc_ctx *ctx, float *rgb);
/* same generator over the scene's full depth range (the reference's "mc" integrator sampler, pathtrace.cpp): an independent
 * estimator used to cross-check the MLT image; result through lmc_direct_read */
int lmc_path_trace(lmc_ctx *ctx, int spp);
/* plain Monte Carlo over GeneratePathBidir samples (path length >= 3), radiance image through lmc_direct_read: a second
 * cross-check estimator that isolates the bidirectional generator from the Markov chain */
int lmc_bidir_mc(lmc_ctx *ctx, int spp);
/* out[0..7] = steps, largeSteps, accepted, gradCalls, cacheQueries, cacheHits, resets, cacheReadyMask; *weight_sum =
 * sum over steps of the splatted weight (film luminance == normalization * weight_sum) */
int lmc_stats(lmc_ctx *ctx, long long *out8, double *weight_sum);
/* per-chain summary, `stride` floats each (>= 32), same layout as the oracle's orc_chain_summary:
 * [valid, camDepth, lightDepth, lsScore, ssScore, scoreSum, time, gaussianInitialized, buffered, sampleIdx,
 *  screenX, screenY, contribR, contribG, contribB, nSplats, pss[0..15]]; which = 0 current, 1 init states.
 * Of an INVALID current state (valid == 0) only lsScore and the technique mean anything: the chain loop reads nothing else of it (mlt.cpp:148,
 * mutation_large.h:87-116), and after an outlier reset (mlt.cpp:151-158) onto an init state that lives on another rank of the job only those
 * are copied -- the path words (time, pss, screen) of such a row are the previous state's. */
int lmc_chain_summary(lmc_ctx *ctx, int which, float *out, int stride);
/* chain relocation (device/relocate.hip; no counterpart in the reference, whose chains are objects a thread walks, mlt.cpp:60-196): from the
 * FIRST step on -- the cache-fill phase included -- the resident chains are kept physically grouped by technique (c,l): after every step's large-step
 * launch, on its stream, the chains that launch gave another technique move to the slots of their technique (the small-step launches' chains are not
 * touched; the step's cache pushes are packed before the move, in chain order).  Invisible in every result except the order of the film's atomics;
 * lmc_chain_summary reports rows in chain order regardless.
 * out4 = [relocations run, chains moved by the last one, adjacent slot pairs whose chains differ in technique key, slots]; -1 when off */
int lmc_relocation_stats(lmc_ctx *ctx, long long *out4);
/* relocations skipped so far because their movers exceeded the staging records (N / 2 after the first full sort): such a step's movers stay where
 * they are -- a performance event only, visible here and as "chains moved by the last one" = 0; -1 when relocation is off */
long long lmc_relocation_skipped(lmc_ctx *ctx);
/* kernel time (ms, HIP events on the launch stream) and launch count of the chain-step kernel since the last call */
int lmc_step_timing(lmc_ctx *ctx, double *kernel_ms, long long *launches);
/* split of the interval the last lmc_step_timing call covered: out3[0] = ms inside the lean small-step kernel
 * (k_step_small, the dominant kernel), out3[1] = ms inside the large-step + generic small-step launches,
 * out3[2] = chain-steps the lean kernel has run since lmc_chains_init (cumulative). */
int lmc_kernel_timing(lmc_ctx *ctx, double *out3);
/* the same interval, the three step launches separately: out4 = [lean small-step ms, large-step ms, generic small-step ms
 * (cache-filling gradient steps; every small step of an H2MC render), cumulative chain-steps of the lean kernel] */
int lmc_kernel_timing_split(lmc_ctx *ctx, double *out4);

/* ---- multi-GPU (one process per GPU; chains sharded by contiguous global id range through lmc_chains_init).
 * The only data-path collective is the sum of the per-GPU films: rank 0 calls lmc_comm_unique_id and the host program
 * ships the 128 bytes to the other ranks (any side channel), every rank calls lmc_comm_init, and after its last step
 * lmc_film_allreduce sums the device films in place with RCCL on the step stream (no host staging) together with the
 * splat-weight sum.  lmc_film_device_ptr exposes the device buffer for hosts that bring their own collective library. */
int lmc_comm_unique_id(unsigned char *out128);
int lmc_comm_init(lmc_ctx *ctx, int n_ranks, int rank, const unsigned char *id128);
int lmc_film_allreduce(lmc_ctx *ctx);
/* Driver plumbing over the job's own communicator (no second communication library): n <= LMC_COMM_MAX_SCALARS host doubles reduced over
 * the ranks in place (op 0 sum, 1 max, 2 min), blocking; and a barrier (own stream drained, then a one-word all-reduce). */
#define LMC_COMM_MAX_SCALARS 64
int lmc_comm_allreduce_f64(lmc_ctx *ctx, double *vals, int n, int op);
int lmc_comm_barrier(lmc_ctx *ctx);
void *lmc_film_device_ptr(lmc_ctx *ctx, long long *n_floats);

/* Batched path program: n evaluations of technique (c,l); SoA, word-major: primary_soa[(2L+1)*n],
 * vert_soa[V*n], grad_soa[2L*n] (word w of item i at [w*n + i]); scene38 as lmc_scene_params.  Host pointers.
 * loglum and grad_soa may each be NULL. */
int lmc_grad_batch(int c, int l, int n, const float *primary_soa, const float *scene38, const float *vert_soa, float *loglum, float *grad_soa);

/* H2MC library (the reference's pathlibbidir.so): symbols evaluate_path_bidir_<c>_<l>_static{,_derv}, the derivative with
 * a 6th `hess` argument ((2L)^2 floats, row i at hess[i*2L]; path.h:122-123, mutation_h2mc.h:74-79), are exported next to
 * the MALA ones.  Batched form: hess_soa[(2L)^2 * n], entry (i,k) of item j at [(i*2L + k)*n + j]. */
int lmc_hess_batch(int c, int l, int n, const float *primary_soa, const float *scene38, const float *vert_soa, float *loglum, float *grad_soa, float *hess_soa);

/* Parity probe: the rows of one global-cache dim as they stand (global_cache.h:21-23 point_cloud_t).  pss: 3000 x dim, weight: 3000,
 * extra: 3000 x 313 = every row's path words then its contribution words, only with `samplecache` (mutation_large_cache.h); any
 * pointer may be NULL.  Returns the number of rows filled, -1 on error. */
int lmc_cache_rows(lmc_ctx *ctx, int dim, float *pss, float *weight, float *extra);
/* Parity probe of LargeStepCache's cache-side pieces on the device (`samplecache`, dim ready): row[i] = sampleCache with the uniform
 * u[i] (global_cache.h:126-137), pdf[i] = evalPdfCache at query[i * dim ..] for technique cl[2 i], cl[2 i + 1] (:139-164).
 * Returns 0, -2 when the dim is not ready or the option is off, -1 on error. */
int lmc_cache_probe(lmc_ctx *ctx, int dim, int n, const float *u, int *row, const float *query, const int *cl, float *pdf);

/* ---- probes used by the parity tests (tests/) ---- */
/* rays: n x [ox,oy,oz,dx,dy,dz,tnear,tfar]; closest hit -> global triangle id (or -1) and t */
int lmc_trace(lmc_ctx *ctx, int n, const float *rays, int *prim, float *t);
int lmc_occluded(lmc_ctx *ctx, int n, const float *rays, int *occluded);
/* mode 0 raw u32, 1 uniform01 bits, 2 one normal_distribution object, 3 mixed (u,u,7 normals per round); + 8: the extension table synthesised from
 * the seed until the stream ticks, as the chain kernels do (device/drng.h) instead of read from memory;
 * out: n_seeds x (n + 66) words, the last 66 = RNG state after the draws [lo, hi, table 64] */
int lmc_rng_probe(int n_seeds, const unsigned long long *seeds, int mode, int n, float mean, float stddev, unsigned *out);
int lmc_kd_probe(int dim, int npts, const float *pts, int nq, const float *q, float radius_sq, int knn, int *out_n, int *out_idx, float *out_dist);
/* test hook: first index of cdf[0..n) that is not below u[i] (the device's search behind the env-map CDF look-ups; = std::lower_bound) */
int lmc_lower_bound_probe(int n, const float *cdf, int nq, const float *u, int *out);
/* measurement aid: ms per launch of a kernel that streams `words` state words per chain in batches of `batch` loads; mode 0 = [word][chain], 1 = [tile of 64][word][lane] */
int lmc_layout_probe(int nChains, int words, int mode, int batch, int reps, double *msPerLaunch);
/* parity probe, evaluated on the device: the deterministic float exp (mode 0) / log (1) / pow (2) of the glossy BSDFs (device/dtrans.h), sin (3) / cos (4) /
 * acos (5) / atan2(x, y) (6) of the sampling code (device/dtrig.h), and glibc's logf as restated for the normal distribution (7, device/drng.h) */
int lmc_trans_probe(int n, int mode, const float *x, const float *y, float *out);
/* measurement hook (LMC_PROF=1): wave cycles per region of the lean small-step kernel since the last call: out16[0 .. LMC_PROF_REGIONS-1]
 * = cycle sums of the regions (dsmall.h PR_*), out16[LMC_PROF_REGIONS] = number of waves */
#define LMC_PROF_REGIONS 15
int lmc_prof_read(lmc_ctx *ctx, unsigned long long *out16);
/* test hook: compares the device-built existence-test grid of one cache dim with the host build; number of differing cells (0 = same), -2 = that cache is not ready */
int lmc_cache_grid_check(lmc_ctx *ctx, int dim);
/* the lean kernel's existence test in front of the cache query, evaluated on the host: out[i] = 1 iff a cache point lies within the radius */
int lmc_cache_filter_probe(int dim, int npts, const float *pts, int nq, const float *q, int *out);
/* ComputeGaussian (mala.cpp:7-52) + GaussianLogPdf (gaussian.cpp:24-36): out n x (3*dim+2) */
int lmc_gauss_probe(int n, int dim, const float *v1, const float *M, float ss, float shk, const float *sc, const float *offset, float *out);
/* Parity probes of the H2MC step's own launches (device/h2hess.hip, h2gauss.hip; reference: the evaluate_path_bidir_<c>_<l>_static_derv
 * programs with `hess`, h2mc.cpp:3-142, gaussian.cpp:24-36), item-major arrays:
 *   lmc_h2_hess_probe   n states of technique (c,l): primary n x (2L+1), vert n x V -> grad n x 16, hess n x 256 (row i at [i * dim], only
 *                       the triangle Eigen reads -- the UPPER triangle of the rows as delivered -- is written, the rest is 0), loglum n
 *   lmc_h2_gauss_probe  n gradients (n x 16) + Hessians (n x dim x dim, upper triangle read) -> the proposal Gaussian of each:
 *                       gauss n x 544 = [mean 16 | logDet | kind (0 dense, 1 isotropic early-out) | .. | covL at 32 (dim x dim) | invCov at 288],
 *                       and, with offset (n x dim) != NULL, px[n] = log N(-offset; mean, invCov^-1) */
int lmc_h2_hess_probe(int c, int l, int n, const float *primary, const float *scene38, const float *vert, float *loglum, float *grad, float *hess);
int lmc_h2_gauss_probe(int n, int dim, const float *grad, const float *hess, float sigma, const float *offset, float *gauss, float *px);
/* HBM counter calibration: `reps` launches of a kernel that reads n_words floats and writes n_words floats with the
 * chain state's access pattern (4 B per lane, SoA, unit stride).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
 * to obtain the bytes-per-count factors profiles/pmc_step_kernel.json applies.  Returns 0 on success. */
int lmc_stream_probe(long long n_words, int reps);

#ifdef __cplusplus
}
#endif
#endif
