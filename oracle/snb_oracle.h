/*
 * snb_oracle.h -- CPU ORACLE for the SlicedNonbondedForce hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (openmm-nonbonded-slicing_amd/csrc -> libsnb_hip.so) never calls into it.
 *
 * It is a plain-C restatement (ours, written from the reference's documented
 * behaviour) of the Reference-platform arithmetic of craabreu/openmm-nonbonded-slicing:
 *   platforms/reference/src/ReferenceNonbondedSlicingKernels.cpp:187-268,339-391  (orchestration)
 *   platforms/reference/src/ReferenceSlicedLJCoulombIxn.cpp:179-507,528-631       (pair loop, Ewald)
 *   platforms/reference/src/ReferenceSlicedLJCoulomb14.cpp:61-95                  (1-4 exceptions)
 *   platforms/reference/src/ReferencePME.cpp:88-183,196-256,264-317,320-396,
 *                                            400-496,499-595,598-702              (sliced PME)
 *   openmmapi/src/SlicedNonbondedForceImpl.cpp:150-185,263-354                    (dispersion corr.)
 * Third-party pieces absent from /root/reference (OpenMM 8.3: ReferenceForce::getDeltaR[Periodic],
 * computeNeighborListVoxelHash, ONE_4PI_EPS0, EPSILON0) are restated from their published
 * behaviour; see DESIGN.md.
 *
 * Parity pinning: tests/test_oracle_kat.py checks this oracle against every closed-form
 * known-answer test the reference's own test-suite holds for this path
 * (tests/TestSlicedNonbondedForce.h:87-135,137-356,358-492,614-681,760-813,883-985).
 * The reference itself cannot be compiled in the build container without stand-in OpenMM
 * headers (OpenMM is not installed), so there is no oracle/_ref build.
 */
#ifndef SNB_ORACLE_H_
#define SNB_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Values as NonbondedSlicing::CalcSlicedNonbondedForceKernel::NonbondedMethod
 * (openmmapi/include/NonbondedSlicingKernels.h:29-36). */
enum { ORC_NoCutoff = 0, ORC_CutoffNonPeriodic = 1, ORC_CutoffPeriodic = 2, ORC_Ewald = 3, ORC_PME = 4, ORC_LJPME = 5 };

typedef struct {
    int    n_atoms;
    int    n_subsets;
    int    method;
    double cutoff;
    int    use_switch;
    double switch_distance;
    double rf_dielectric;
    double alpha;            /* Ewald / PME separation parameter                  */
    int    grid[3];          /* PME mesh                                          */
    int    kmax[3];          /* Ewald: number of k vectors per axis (numRx,y,z)   */
    double alpha_d;          /* LJPME dispersion separation parameter             */
    int    dgrid[3];         /* LJPME dispersion mesh                             */
    int    exceptions_periodic;
    int    use_dispersion_correction;
    int    include_direct;
    int    include_reciprocal;
    int    background_term;  /* 1 = OpenMM >= 8.3 neutralising background (Q3)    */
    int    correct_q1;       /* 1 = mathematically correct subset stride in PME force interpolation;
                                0 = reproduce the reference's ngrid[2] stride (Quirk Q1; identical when nx == nz) */
} orc_config;

/* One evaluation.  All arrays are caller-owned.
 *   pos[N][3], box[9] (rows a,b,c; lower triangular), charge/sigma/epsilon[N] raw particle parameters,
 *   subset[N], exceptions: n_exc pairs exc_pairs[n_exc][2] with (chargeProd, sigma, epsilon),
 *   lambdas[S][2] (Coul, vdW), dispersion coefficients computed internally when requested.
 * Outputs: forces[N][3] (ADDED to, like the reference), slice_energies[S][2] raw (overwritten).
 * Returns 0, or a negative error code (-1: box smaller than 2*cutoff, -2: bad argument). */
int orc_evaluate(const orc_config* cfg,
                 const double* pos, const double* box,
                 const double* charge, const double* sigma, const double* epsilon, const int* subset,
                 int n_exc, const int* exc_pairs, const double* exc_chargeprod, const double* exc_sigma, const double* exc_epsilon,
                 const double* lambdas, const double* disp_coef /* [S] or NULL */,
                 double* forces, double* slice_energies);

/* Per-slice long-range dispersion-correction coefficients (SlicedNonbondedForceImpl.cpp:263-354).
 * out[S]; returns 0. */
int orc_dispersion_coefficients(int n_atoms, int n_subsets, const double* sigma, const double* epsilon, const int* subset,
                                double cutoff, int use_switch, double switch_distance, double* out);

/* B-spline moduli of order `order` for an axis of n points (ReferencePME.cpp:88-183). */
void orc_bspline_moduli(int n, int order, double* out);

/* Unnormalised complex 3D FFT in place, data[nx][ny][nz] interleaved re/im. sign=-1 forward, +1 backward. */
void orc_fft3d(double* data, int nx, int ny, int nz, int sign);

/* DIAGNOSTIC for the parity tests (not part of the restated path): the non-excluded pairs with |r^2/cutoff^2 - 1| < rel_eps, i.e.
 * the pairs a single-precision evaluation of r^2 may put on the other side of the (discontinuous) truncation.  out_ij[max_out][2],
 * out_vals[max_out][4] = r, |F| the pair puts on either atom, raw Coulomb energy, raw vdW energy.  Returns the pair count. */
long long orc_cutoff_band_pairs(const orc_config* cfg, const double* pos, const double* box,
                                const double* charge, const double* sigma, const double* epsilon, const int* subset,
                                int n_exc, const int* exc_pairs, const double* lambdas, double rel_eps,
                                long long max_out, int* out_ij, double* out_vals);

/* Number of within-cutoff, non-excluded pairs found by the last orc_evaluate() in this thread's process. */
long long orc_last_pair_count(void);

/* OpenMP threads used by the PME sections (the pair loop is serial like the reference's). */
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif

#endif
