/*
 * snb.h -- C ABI of the MI355X-native SlicedNonbondedForce engine (libsnb_hip.so).
 *
 * This is the drop-in boundary.  It replaces, for the force/energy hot path, what the reference puts
 * behind  NonbondedSlicing::CalcSlicedNonbondedForceKernel
 *   (openmmapi/include/NonbondedSlicingKernels.h:27-85):
 *     initialize(system, force)                      :48   -> snb_create + snb_set_particles/_exceptions/_lambdas/...
 *     execute(context, forces, energy, direct, recip):59   -> snb_set_box/_positions + snb_execute + snb_get_forces
 *     copyParametersToContext(context, force)        :66   -> snb_set_particles/_exceptions/_dispersion_coefficients
 *     getPMEParameters(alpha,nx,ny,nz)               :75   -> snb_get_pme_parameters
 *     getLJPMEParameters(alpha,nx,ny,nz)             :84   -> snb_get_ljpme_parameters
 * and the device work the reference enqueues from CommonCalcSlicedNonbondedForceKernel::execute
 *   (platforms/common/src/CommonNonbondedSlicingKernels.cpp:846-1402).
 *
 * Plain pointers and sizes only: no C++ types, no torch types.  Caller owns every host array;
 * the engine owns all device memory.  One handle = one device = one HIP stream as far as the caller can see: all work is ordered on
 * snb_config.stream (replayed steps fork a part of it onto an internal second stream and join it before they end); a handle is not
 * thread-safe.  Errors are status codes; snb_last_error() gives the message an adapter rethrows as
 * OpenMM::OpenMMException (INTEGRATION.md shows the adapter).
 */
#ifndef SNB_H_
#define SNB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNB_ABI_VERSION 6

typedef struct snb_engine* snb_handle;

typedef enum {
    SNB_OK = 0,
    SNB_ERR_INVALID_ARGUMENT = 1,
    SNB_ERR_HIP = 2,              /* a HIP runtime call failed (no GPU, out of memory, ...)          */
    SNB_ERR_BOX_TOO_SMALL = 3,    /* "The periodic box size has decreased to less than twice the nonbonded cutoff."
                                     (ReferenceNonbondedSlicingKernels.cpp:202-204)                     */
    SNB_ERR_NOT_PME = 4,          /* getPMEParametersInContext on a non-PME engine (CommonNonbondedSlicingKernels.cpp:1571-1581) */
    SNB_ERR_STATE = 5,            /* call order violated (e.g. execute before positions were set)     */
    SNB_ERR_UNSUPPORTED = 6
} snb_status;

/* Values of CalcSlicedNonbondedForceKernel::NonbondedMethod (NonbondedSlicingKernels.h:29-36). */
typedef enum {
    SNB_NoCutoff = 0, SNB_CutoffNonPeriodic = 1, SNB_CutoffPeriodic = 2, SNB_Ewald = 3, SNB_PME = 4, SNB_LJPME = 5
} snb_method;

/* The precision modes of the reference's GPU platforms (CudaPlatform "Precision" property: single / mixed / double).
 * SNB_MIXED: single-precision pair and mesh arithmetic as SNB_SINGLE, but the direct-space forces are accumulated in 64-bit fixed
 * point (2^32 per kJ/mol/nm) the way the reference's GPU platforms accumulate every force (platforms/common/src/kernels/pme.cc:381-389,
 * realToFixedPoint; CommonNonbondedSlicingKernels.cpp getLongForceBuffer): integer sums are independent of the order in which waves
 * finish, so with the charge mesh already spread in fixed point the force of a step is reproducible bit for bit; the accumulators and
 * the reciprocal part are summed in double on delivery.  Range: |force component| < 2^31 kJ/mol/nm (saturating), as in the reference. */
typedef enum { SNB_SINGLE = 0, SNB_DOUBLE = 1, SNB_MIXED = 2 } snb_precision;

typedef struct {
    int32_t abi_version;        /* SNB_ABI_VERSION                                                      */
    int32_t n_atoms;
    int32_t n_subsets;
    int32_t method;             /* snb_method                                                           */
    int32_t precision;          /* snb_precision: arithmetic type of every kernel                       */
    int32_t use_switch;         /* LJ switching function (ignored for NoCutoff and LJPME, Quirk Q2)     */
    int32_t exceptions_periodic;
    int32_t device;             /* HIP device ordinal                                                   */
    double  cutoff;
    double  switch_distance;
    double  rf_dielectric;
    double  alpha;              /* Ewald/PME separation parameter (explicit; > 0 for Ewald/PME/LJPME)   */
    int32_t grid[3];            /* requested PME mesh; rounded up to the next size the FFT supports     */
    int32_t kmax[3];            /* Ewald: numRx, numRy, numRz (ReferenceSlicedLJCoulombIxn.cpp:115-121) */
    double  alpha_d;            /* LJPME dispersion separation parameter                                */
    int32_t dgrid[3];           /* LJPME dispersion mesh                                                */
    double  neighbor_padding;   /* nm added to the cutoff when building tiles (0 = rebuild every call)  */
    int32_t rebuild_interval;   /* rebuild tiles every this many executes (0, 1: every execute); < 0: automatic -- when an atom has
                                 * moved 0.8 * neighbor_padding / 2 since the last rebuild, and after -rebuild_interval executes at the latest.
                                 * (With a fixed interval > 4 the next list is built three executes ahead on an internal stream, from a
                                 * copy of the positions, and comes into use when the rebuild falls due: it is then three executes
                                 * older than one built in line.) */
    int32_t shard_rank;         /* multi-GPU: this engine owns PME subsets J with J % shard_count == shard_rank */
    int32_t shard_count;        /*            and direct-space work items w with w % shard_count == shard_rank; 1 = unsharded */
    int32_t disable_graph;      /* 1 = enqueue every step eagerly (default 0: forces-only steps replay a captured hipGraph)  */
    int32_t host_neighbor_build; /* 1 = always build the tile lists on the host (default 0: on the GPU when the box allows it)            */
    void*   stream;             /* hipStream_t to enqueue on, or NULL for an engine-owned stream        */
} snb_config;

/* Aggregate counters for measurement (bench.py / DESIGN.md byte model). */
typedef struct {
    int64_t n_tiles;            /* 32x32 tiles processed per evaluation by this engine (T)             */
    int64_t n_blocks;           /* 32-atom blocks (padded atoms / 32)                                   */
    int64_t n_padded_atoms;
    int64_t n_exclusion_tiles;  /* tiles that carry an exclusion mask                                   */
    int64_t n_exclusions;       /* excluded pairs (exceptions)                                          */
    int64_t n_14;               /* exceptions with non-zero parameters (Q6)                             */
    int64_t n_rebuilds;
    int32_t grid[3];            /* PME mesh actually used                                               */
    int32_t dgrid[3];
    double  last_direct_ms;     /* HIP-event time of the last direct-space pair kernel launch           */
    double  last_recip_ms;      /* HIP-event time of the last reciprocal pipeline (all its kernels)     */
    double  last_total_ms;      /* HIP-event time of the last snb_execute (all kernels)                 */
    double  last_rebuild_ms;    /* wall time of the last neighbour rebuild                              */
    double  sum_direct_ms;      /* cumulative HIP-event times since snb_reset_timers (harvested lazily,   */
    double  sum_recip_ms;       /*   no per-step host synchronisation)                                  */
    double  sum_total_ms;
    int64_t n_timed;            /* executes included in the sums                                        */
    int64_t n_host_rebuilds;    /* rebuilds that fell back to the host builder (triclinic / non-periodic / tiny boxes) */
    int64_t n_list_overruns;    /* list lifetimes in which an atom moved more than neighbor_padding / 2 (pairs may have been missed:
                                 * shorten rebuild_interval, widen the padding, or use the automatic mode)                */
    /* (ABI 6) per-kernel begin/end stamps of the timed (eager) steps, cumulative since snb_reset_timers.  Slots (SNB_K_*): 0 position
     * gather, 1 charge spreading (+ fused forward z FFT), 2 forward z FFT when separate, 3 forward y FFT, 4 x FFT + slice energies +
     * lambda mix + inverse x FFT, 5 inverse y FFT, 6 inverse z FFT, 7 force interpolation (+ user-order force write); 8..15 the same
     * for the LJPME dispersion mesh (8 unused). */
    double  sum_kernel_ms[16];
    int64_t n_kernel_timed[16];
    int64_t n_spread_strays;    /* atoms of the LAST execute whose spreading footprint had left their work-group's LDS region (drift beyond the
                                 * margin since the last re-sort): handled one by one, exactly, by the merge kernel -- normally 0 */
} snb_stats;
#define SNB_K_GATHER 0
#define SNB_K_SPREAD 1
#define SNB_K_FFT_Z_FWD 2
#define SNB_K_FFT_Y_FWD 3
#define SNB_K_CONVOLVE_X 4
#define SNB_K_FFT_Y_INV 5
#define SNB_K_FFT_Z_INV 6
#define SNB_K_INTERPOLATE 7
#define SNB_K_DISPERSION_MESH 8

/* -- lifetime ---------------------------------------------------------------------------------- */
snb_status snb_create(const snb_config* cfg, snb_handle* out);
void       snb_destroy(snb_handle h);
const char* snb_last_error(snb_handle h);        /* h may be NULL: error of the last failed snb_create */

/* -- parameters (host arrays; effective values, i.e. after parameter offsets) ------------------ */
/* charge[N], sigma[N], epsilon[N] as NonbondedForce::getParticleParameters; subset[N] in [0,n_subsets). */
snb_status snb_set_particles(snb_handle h, const double* charge, const double* sigma, const double* epsilon, const int32_t* subset);
/* All exceptions: pairs[m][2], chargeProd[m], sigma[m], epsilon[m].  Every exception is an exclusion; those with
 * chargeProd != 0 or epsilon != 0 or force14[k] != 0 are also 1-4 interactions (Quirk Q6,
 * ReferenceNonbondedSlicingKernels.cpp:99-112).  force14 may be NULL. */
snb_status snb_set_exceptions(snb_handle h, int32_t m, const int32_t* pairs, const double* charge_prod, const double* sigma,
                              const double* epsilon, const int32_t* force14);
/* Parameter offsets (SlicedNonbondedForce::addParticleParameterOffset / addExceptionParameterOffset), applied ON THE DEVICE as the
 * reference does (platforms/common/src/kernels/nonbondedParameters.cc:4-179): snb_set_particles / snb_set_exceptions give the BASE
 * values and   effective = base + sum_k global[k] * delta_k .   particle_delta[n][3] = (charge, sigma, epsilon) per unit of the global
 * parameter particle_global[n]; exception_delta[n][3] = (chargeProd, sigma, epsilon); an exception that carries an offset counts as
 * a 1-4 interaction whatever its base values (Q6).  Call after snb_set_exceptions; globals start at 0. */
snb_status snb_set_parameter_offsets(snb_handle h, int32_t n_globals,
                                     int32_t n_particle_offsets, const int32_t* particle, const int32_t* particle_global, const double* particle_delta,
                                     int32_t n_exception_offsets, const int32_t* exception, const int32_t* exception_global, const double* exception_delta);
/* New values of the global parameters the offsets refer to (Context::setParameter): the next snb_execute recomputes the effective
 * parameters with two small kernels -- no re-sort, no tile rebuild, no graph re-capture, no host synchronisation. */
snb_status snb_set_global_parameters(snb_handle h, int32_t n_globals, const double* values);
/* mask[S]: the slices whose raw energies a DERIVATIVE-ONLY step (snb_execute with include_energy == 2) must produce -- the slices bound
 * to a global parameter whose derivative was requested.  The reference adds dE/dlambda on every execute
 * (CommonNonbondedSlicingKernels.cpp:712-718) but needs the energy of those slices only; here the pair kernel runs its forces-only
 * arithmetic on the tiles of every other slice (on the 300k-atom box 95 % of the tiles are solvent-solvent), the reciprocal kernel
 * skips their Gram sums.  Default: every slice.  The energies of slices outside the mask are unspecified after such a step. */
snb_status snb_set_energy_slices(snb_handle h, const int32_t* mask);
/* lambdas[S][2] = (Coulomb, vdW) per slice, S = n(n+1)/2, slice(i,j) = max(max+1)/2+min (SlicedNonbondedForce.h:22). */
snb_status snb_set_lambdas(snb_handle h, const double* lambdas);
/* Per-slice dispersion-correction coefficients (SlicedNonbondedForceImpl.cpp:263-354); NULL = none. */
snb_status snb_set_dispersion_coefficients(snb_handle h, const double* coef);
/* Host helper (no GPU work): the coefficients themselves, restating calcDispersionCorrections. out[S]. */
snb_status snb_compute_dispersion_coefficients(int32_t n_atoms, int32_t n_subsets, const double* sigma, const double* epsilon,
                                               const int32_t* subset, double cutoff, int32_t use_switch, double switch_distance,
                                               double* out);

/* -- per-step state ---------------------------------------------------------------------------- */
snb_status snb_set_box(snb_handle h, const double* box9);   /* rows a, b, c (reduced, lower triangular) */
/* pos: [N][3] (is_double=1: double, else float) or, when stride4 != 0, [N][4] (OpenMM posq layout).  is_device != 0:
 * a device pointer valid on the engine's device. */
snb_status snb_set_positions(snb_handle h, const void* pos, int32_t is_device, int32_t is_double, int32_t stride4);
snb_status snb_rebuild_neighbors(snb_handle h);             /* force a tile rebuild at the next execute */

/* -- the hot path ------------------------------------------------------------------------------ */
/* Enqueues one evaluation; forces stay on the device until snb_get_forces.  include_energy != 0: the raw per-slice energies are
 * accumulated as well (the step of every force with energy-parameter derivatives: the reference adds dE/dlambda on every execute,
 * CommonNonbondedSlicingKernels.cpp:712-718) and summed ON THE DEVICE as the step's last kernel.  With energy == NULL nothing is read
 * back and nothing synchronises -- such a step replays a captured graph like a forces-only one; snb_get_slice_energies fetches the
 * sums when the caller needs them.  With energy != NULL it receives sum_slices lambda*E (this synchronises the stream).
 * include_energy == 2: a derivative-only step -- as 1, restricted to the slices of snb_set_energy_slices (energy must be NULL). */
snb_status snb_execute(snb_handle h, int32_t include_forces, int32_t include_energy, int32_t include_direct,
                       int32_t include_reciprocal, double* energy);
/* out: [N][3] in the type selected by is_double; accumulate != 0 adds to what is there (the reference
 * accumulates into the platform's force buffer). */
snb_status snb_get_forces(snb_handle h, void* out, int32_t is_device, int32_t is_double, int32_t accumulate);
/* Optional: name the device buffer ([N][3], type by is_double) that every following snb_execute writes (accumulate == 0) or adds
 * (accumulate != 0) the forces to as the last kernel of the step -- it then belongs to the replayed step graph and a later
 * snb_get_forces on the same pointer is a no-op.  out == NULL switches this off.  (The reference's execute() adds into the
 * platform's force buffer itself, NonbondedSlicingKernels.h:59; this is that behaviour without an extra launch per step.) */
snb_status snb_set_force_output(snb_handle h, void* out, int32_t is_double, int32_t accumulate);
/* Optional, sharded engines: direct-space ownership of the 32-atom i-blocks.  The engine evaluates the tiles of block I when
 * I % period lies in [begin, end); the default is (shard_rank, shard_rank + 1, shard_count).  The ranges of all ranks must partition
 * [0, period).  Lets the host shift direct-space work away from ranks that carry PME grids (the role of the load balancing between
 * devices in OpenMM's parallel kernels, which the reference inherits from its platform); takes effect with a neighbour rebuild at
 * the next snb_execute.  An empty range (begin == end) is allowed. */
snb_status snb_set_shard_blocks(snb_handle h, int32_t begin, int32_t end, int32_t period);
/* Raw (unscaled) energies of the last execute with include_energy: out[S][2] = (Coulomb, vdW). */
snb_status snb_get_slice_energies(snb_handle h, double* out);
/* Device address of those energies: double[S][2], complete (pair sums, reciprocal sums, self, background and dispersion-correction
 * terms) once the last kernel of an energy step has run on the engine's stream.  For a caller that accumulates dE/dlambda on the
 * device itself (OpenMM's energy-parameter-derivative buffer): no read-back, no synchronisation.  The address is fixed for the
 * engine's lifetime. */
snb_status snb_slice_energies_device(snb_handle h, const double** out);
snb_status snb_synchronize(snb_handle h);

/* -- queries ----------------------------------------------------------------------------------- */
snb_status snb_get_pme_parameters(snb_handle h, double* alpha, int32_t grid[3]);
snb_status snb_get_ljpme_parameters(snb_handle h, double* alpha, int32_t grid[3]);
snb_status snb_get_stats(snb_handle h, snb_stats* out);
snb_status snb_reset_timers(snb_handle h);
/* Every n-th execute is enqueued as plain launches with HIP-event stamps around the pair kernel and the reciprocal pipeline (the
 * samples behind snb_stats' kernel timers); the others replay the captured step graph.  Default 32; a short measured region asks
 * for more samples.  n <= 0: never (no timers). */
snb_status snb_set_timing_interval(snb_handle h, int32_t n);
/* Smallest FFT-legal mesh size >= n: no prime factor above 13, the rule of the reference's GPU platforms
 * (platforms/common/include/FFT3DFactory.h:31-47); at least 6. */
int32_t    snb_legal_grid_size(int32_t n);
int32_t    snb_abi_version(void);

/* -- unit-test hooks for the reciprocal building blocks (used by tests/ only; they run the same
 *    kernels the pipeline uses) --------------------------------------------------------------- */
/* Batched 3D real-to-complex forward FFT followed by the inverse, through the engine's own FFT kernels:
 * in[batch][nx][ny][nz] (host, double) -> spectrum[batch][nx][ny][nz/2+1][2] and roundtrip[batch][nx][ny][nz]
 * (unnormalised, like the reference's FFT3D: roundtrip = nx*ny*nz * in).  precision selects the kernel type. */
snb_status snb_test_fft3d(int32_t precision, int32_t device, int32_t batch, int32_t nx, int32_t ny, int32_t nz,
                          const double* in, double* spectrum, double* roundtrip);

#ifdef __cplusplus
}
#endif
#endif /* SNB_H_ */
