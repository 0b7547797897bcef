/*
 * radtxfr_hip.h -- C ABI of libradtxfr_hip.so, the MI355X (gfx950) engine behind the
 * reference's Python hot-path functions.
 *
 * The reference (westi024/RadTxfr) is 100 % Python and has no FFI of its own; its boundary for
 * this path is the set of Python call signatures listed in SURVEY.md section 8b. Each entry point
 * below is what a binding for one of those functions calls; the cited lines are the reference
 * code the entry point replaces. radtxfr_amd/_lib.py is the ctypes binding; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; rtx_last_error() gives the text
 *     (thread-local). No C++ exception crosses the ABI.
 *   - pointers are DEVICE pointers unless the name ends in _h (host).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). Calls are asynchronous
 *     with respect to the host and never allocate, free or synchronise, except the
 *     rtx_lines_* / rtx_prep_* create/free calls, which say so.
 *   - spectra are float32, wavenumber-contiguous; a layer-resolved array is layer-major
 *     ([n_layers][ld], element (k,i) at k*ld+i) so loads along the wavenumber axis coalesce.
 *   - the spectral grid is never materialised: X[i] = xmin + (offset+i)*step in fp64 (product
 *     then sum, unfused, as np.linspace builds it) and X[n_total-1] = xmax exactly.
 */
#ifndef RADTXFR_HIP_H
#define RADTXFR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_VERSION 100 /* major*10000 + minor*100 + patch */

/* Uniform spectral axis = np.linspace(xmin, xmax, n_total) as built by
 * radiative_transfer.py:251-271 (make_spectral_axis); [offset, offset+n) is the part this
 * call (this GPU) owns, so a wavenumber shard evaluates bit-identical grid values. */
typedef struct rtx_grid {
  double xmin;
  double xmax;
  double step; /* (xmax-xmin)/(n_total-1), the value np.linspace computes */
  int64_t n_total;
  int64_t offset;
  int64_t n;
} rtx_grid;

int rtx_version(void);
const char* rtx_last_error(void);
/* name[0..len) <- device name, *n_cu <- compute units, of the current HIP device */
int rtx_device_info(char* name_h, int len, int* n_cu_h);

/* ------------------------------------------------------------------------------------------
 * Line table. Replaces LOCAL_TABLE_CACHE[name]['data'] (misc/hapi.py:438-463) for the columns
 * absorptionCoefficient_Voigt reads (misc/hapi.py:11059-11125). Rows must be sorted by nu.
 * `species_h[l]` in [0,n_species) indexes the caller's list of distinct (molec_id,local_iso_id)
 * pairs; the per-species quantities of rtx_line_prep are given in that order.
 * n_self_h / deltap_air_h / delta_self_h may be NULL (hapi's fallbacks apply: n_self->n_air,
 * others 0; misc/hapi.py:11097-11125). Allocates device memory and synchronises. */
typedef struct rtx_lines rtx_lines;
int rtx_lines_create(int64_t n_lines, int n_species, const double* nu_h, const double* sw_h,
                     const double* elower_h, const double* gamma_air_h, const double* gamma_self_h,
                     const double* n_air_h, const double* n_self_h, const double* delta_air_h,
                     const double* deltap_air_h, const double* delta_self_h,
                     const int32_t* species_h, rtx_lines** out);
int rtx_lines_free(rtx_lines* lines);
int64_t rtx_lines_count(const rtx_lines* lines);
/* Optional speed-dependence columns SD_air / SD_self (misc/hapi.py:10884-10887), in the order of the rows given to
 * rtx_lines_create; either may be NULL (= 0). Only rtx_line_prep_profile(RTX_PROFILE_SDVOIGT) reads them. */
int rtx_lines_set_sd(rtx_lines* lines, const double* sd_air_h, const double* sd_self_h);
/* Optional deltap_self column: temperature dependence of the self-induced pressure shift,
 * Shift0 += abun_self * (delta_self + deltap_self*(T - Tref)) * p (misc/hapi.py:11120-11128). NULL = 0. */
int rtx_lines_set_deltap_self(rtx_lines* lines, const double* deltap_self_h);

/* ------------------------------------------------------------------------------------------
 * Per-(line, layer) prologue, fp64. Replaces the per-line environment block of
 * absorptionCoefficient_Voigt, misc/hapi.py:11068-11134: S(T) (:10169-10175), GammaD (:11085-
 * 11087), Gamma0 (:10183-10184), Shift0 (:11127-11128), OmegaWingF (:11131) and the two
 * bisect() window bounds (:11133-11134), for every line and every layer at once.
 *
 * Host inputs (small, copied with hipMemcpyAsync on `stream`; kept alive by the prep object):
 *   T_h[n_layers] K, p_atm_h[n_layers] atm
 *   qratio_h[n_species*n_layers]   Q(Tref)/Q(T_k) per species (TIPS, misc/hapi.py:11069-11070)
 *   weight_h[n_species*n_layers]   factor multiplying S(T) for that species in that layer:
 *       hapi path : factor/natural_abundance*abundance  (misc/hapi.py:11136-11137)
 *       OD path   : volumeConcentration(p,T)*x_m*PL*1e5  (SURVEY 8(a-3)); 0 drops the species
 *   mass_h[n_species]  g/mol (misc/hapi.py:11086)
 *   dil_air, dil_self  Diluent fractions (misc/hapi.py:11025-11032)
 *   omega_wing, omega_wing_hw  (misc/hapi.py:11131); intensity_threshold (:11082)
 *   scale  power of two folded into the fp32 strengths (HITRAN-unit cross sections are ~1e-25)
 * rtx_prep_create allocates (n_lines*max_layers records, plus per-tile line ranges for grids of up
 * to max_points points) and may synchronise; rtx_line_prep only enqueues work. One prep object
 * can be re-used for any atmospheric state / grid within its capacity. */
typedef struct rtx_prep rtx_prep;
int rtx_prep_create(const rtx_lines* lines, int max_layers, int64_t max_points, rtx_prep** out);
int rtx_prep_free(rtx_prep* prep);
int rtx_line_prep(rtx_prep* prep, const rtx_lines* lines, const rtx_grid* grid, int n_layers,
                  const double* T_h, const double* p_atm_h, const double* qratio_h,
                  const double* weight_h, const double* mass_h, double dil_air, double dil_self,
                  double omega_wing, double omega_wing_hw, double intensity_threshold,
                  double scale, void* stream);
/* The same prologue for the other hapi profiles that share it; rtx_voigt_sum then sums whatever
 * profile the records describe. rtx_line_prep == profile RTX_PROFILE_VOIGT.
 *   RTX_PROFILE_LORENTZ  absorptionCoefficient_Lorentz, misc/hapi.py:11144-11375: PROFILE_LORENTZ (:10150),
 *                        OmegaWingF = max(OmegaWing, OmegaWingHW*Gamma0) (:11364)
 *   RTX_PROFILE_DOPPLER  absorptionCoefficient_Doppler, misc/hapi.py:11384-11559: PROFILE_DOPPLER (:10160), its own
 *                        GammaD constants (:11534-11538), OmegaWingF = max(OmegaWing, OmegaWingHW*GammaD) (:11540),
 *                        Shift0 = dil_air * delta_air * p (:11543; pass dil_air = 0 for LineShift=False) */
#define RTX_PROFILE_VOIGT 0
#define RTX_PROFILE_LORENTZ 1
#define RTX_PROFILE_DOPPLER 2
#define RTX_PROFILE_SDVOIGT 3 /* records for rtx_sdvoigt_sum (below); windows as for Voigt */
/* Hot tiles (line lists cluster: a band head puts thousands of candidate lines on one line-sum tile, and one workgroup per
 * tile would serialise the launch). The prologue bounds, on the host, the candidates any tile of `grid` can have -- no
 * window is wider than max(OmegaWing, OmegaWingHW * gamma, misc/hapi.py:11131) with the table's column extremes -- and
 * rtx_voigt_sum cuts tiles with more than 768 candidates into parts of 256 evaluated by extra workgroups and summed in a fixed
 * order (results stay bit-reproducible and independent of wavenumber sharding). This returns the number of extra parts
 * the last prologue allowed for: 0 = no tile of that table / grid can be hot and the extra kernels are never launched.
 * A bound that grows (new table or grid, much wider wings) re-allocates the part workspace inside rtx_line_prep*: the one
 * case in which a prologue call allocates and synchronises. */
int64_t rtx_prep_split_bound(const rtx_prep* prep);
int rtx_line_prep_profile(rtx_prep* prep, const rtx_lines* lines, const rtx_grid* grid, int n_layers,
                          const double* T_h, const double* p_atm_h, const double* qratio_h,
                          const double* weight_h, const double* mass_h, double dil_air,
                          double dil_self, double omega_wing, double omega_wing_hw,
                          double intensity_threshold, double scale, int profile, void* stream);

/* ------------------------------------------------------------------------------------------
 * Voigt line-sum. Replaces the per-line PROFILE_VOIGT + scatter-add loop, misc/hapi.py:11050,
 * 11135-11138 (PROFILE_VOIGT :10131 -> pcqsdhc PART1 :9900-9915 -> hum1_wei :9833-9844) with a
 * gather over the lines whose window covers each grid point.
 *   out_f32[n_layers][ld]  (may be NULL)   sum * 1            -> layer optical depths / k(nu)
 *   out_f64[n_layers][ld]  (may be NULL)   (double)sum/scale  -> hapi Xsect
 * compute_OD contract: radiative_transfer.py:395-456. */
/* points per line-sum workgroup tile (capacity granularity of rtx_prep_create) */
int rtx_voigt_tile_points(void);
int rtx_voigt_sum(const rtx_prep* prep, const rtx_grid* grid, int n_layers, float* out_f32,
                  double* out_f64, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Planck radiance. Replaces planckian(), radiative_transfer.py:792-848.
 *   X == NULL : spectral axis = grid; else X[nx] (fp64 wavenumbers or micrometres)
 *   out[nx][nT] float64 (C order, spectral axis first, as the reference returns it) */
int rtx_planck(const rtx_grid* grid, const double* X, int64_t nx, const double* T, int64_t nT,
               int wavelength, double* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Brightness temperature and its inverse. Replace brightnessTemperature(),
 * radiative_transfer.py:851-933, and BT2L(), :936-1014 (fp64; bad_value where the reference masks:
 * non-finite or non-positive radiance :922-923, non-finite radiance or T<=0 :1004-1005).
 *   X[nx] fp64; in/out [nx][m] fp64, spectral axis first. */
int rtx_brightness_temperature(const double* X, int64_t nx, const double* L, int64_t m,
                               int wavelength, double bad_value, double* T_out, void* stream);
int rtx_bt2l(const double* X, int64_t nx, const double* T, int64_t m, int wavelength,
             double bad_value, double* L_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * TUD integration. Replaces the body of compute_TUD after the OD loop,
 * radiative_transfer.py:340-392: on-the-fly Planck, tau, upwelling recurrence (:346-356),
 * N_angle-stream downwelling recurrence and its cos*sin average (:368-389).
 *   OD[n_layers][ld] float32;  T_h[n_layers];
 *   n_alt sensor altitudes: mask_h[a*n_layers+k] = (Z[k] <= zs[a]) (:348), count = popcount;
 *   n_mu slant factors mu_h (:313);  n_down = layers the downwelling loop covers (quirk: the
 *   count of the LAST altitude, :353,370);
 *   outputs: tau[n_alt*n_mu][ld_out], Lu[n_alt*n_mu][ld_out] (index a*n_mu+m), Ld[ld_out];
 *   Ld_angles: NULL, or [n_angle][ld_out] per-stream downwelling radiances (what opts['save']
 *   dumps as Ld, :374-386);
 *   return_od != 0 puts sum(OD*mu) in the tau slot (:349-350).
 * The weighted stream sum is evaluated as sum_k B_k [G(S_k) - G(S_k+1)], G(S) = sum_q w_q exp(-S/cos
 * theta_q) tabulated per n_angle in fp64 (the same quadrature, summed over the angles first); with
 * Ld_angles the streams themselves are run (both against the reference's recurrence to <= 1e-5). */
int rtx_tud(const float* OD, int64_t ld, const rtx_grid* grid, int n_layers, const double* T_h,
            int n_alt, const uint8_t* mask_h, int n_mu, const double* mu_h, int n_down,
            int n_angle, int return_od, float* tau, float* Lu, float* Ld, float* Ld_angles,
            int64_t ld_out, void* stream);
/* The tabulated G of rtx_tud, for host-side checks (no device involved): rtx_tud_gtable_size() doubles, rows of
 * {interval centre, a0 .. a6}: G(S) = sum a_k (S - centre)^k on the row's interval; intervals: 16 per binade of
 * S + 2^-6 below S = 16 (row = (bits(float(S) + 2^-6) >> 19) - (bits(2^-6) >> 19)), width 1/2 from there to S = 48.
 * g0_h receives G(0) = the sum of the quadrature weights cos(theta) sin(theta) (radiative_transfer.py:387). */
int rtx_tud_gtable_size(void);
int rtx_tud_gtable(int n_angle, double* table_h, double* g0_h);

/* compute_TUD in ONE call: radiative_transfer.py:274-392 with compute_OD (:395-456, the LBLRTM run) replaced by the
 * Voigt line-sum -- rtx_line_prep (profile Voigt, scale 1) + rtx_voigt_sum into OD[n_layers][ld_od] + rtx_tud, enqueued
 * back to back on `stream`. Arguments as for those three entry points; OD is caller-owned (it is also an output: the
 * reference's returnOD / save options expose it). One FFI crossing per atmosphere instead of three. */
int rtx_compute_tud(rtx_prep* prep, const rtx_lines* lines, const rtx_grid* grid, int n_layers,
                    const double* T_h, const double* p_atm_h, const double* qratio_h,
                    const double* weight_h, const double* mass_h, double dil_air, double dil_self,
                    double omega_wing, double omega_wing_hw, double intensity_threshold, int n_alt,
                    const uint8_t* mask_h, int n_mu, const double* mu_h, int n_down, int n_angle,
                    int return_od, float* OD, int64_t ld_od, float* tau, float* Lu, float* Ld,
                    int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * At-sensor radiance. Replaces compute_LWIR_apparent_radiance(), radiative_transfer.py:1017-
 * 1069: L = tau*(emis*B(Ts+dT) + (1-emis)*Ld) + La.
 *   X[nX] fp64; emis[nX][nE]; Ts[nA] fp64; tau,La,Ld [nX][nA]; dT[nT] fp64 or NULL (nT=0)
 *   L [nX][nE][nA][max(nT,1)], Ls same shape or NULL. All spectra float32 (the reference's
 *   own caller casts to float32 first: Compute_LWIR_Apparent_Radiance.py:9-20). */
int rtx_apparent_radiance(const double* X, int64_t nX, const float* emis, int64_t nE,
                          const double* Ts, int64_t nA, const float* tau, const float* La,
                          const float* Ld, const double* dT, int64_t nT, float* L, float* Ls,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * MAKO instrument line shape. Replaces ILS_MAKO (triangle), radiative_transfer.py:1236-1256
 * (kind 0) and the Gaussian ILS_MAKO.py:21-33 (kind 1): Y_out[b][s] = sum_i w_b(X_i) Y[i][s] /
 * sum_i w_b(X_i). Band centres/widths are computed by the caller (host, :1226-1241) and passed:
 *   centre[nB] (= scale*X_out+shift), sigma[nB]: fp64 DEVICE arrays;
 *   Y [nx][nS] float32 (spectral axis first, as the reference lays it out), ldY = row stride;
 *   Y_out [nB][nS] float32. X == NULL -> uniform grid, else explicit fp64 axis (ascending). */
int rtx_ils(int kind, const rtx_grid* grid, const double* X, int64_t nx, const float* Y,
            int64_t nS, int64_t ldY, int nB, const double* centre, const double* sigma,
            float* Y_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Knot spectra -> monochromatic axis, column-wise np.interp. The reference's emissivity databases
 * live on ~1 cm^-1 knots (Generate_ASTER_emissivity_DB.py:48-52,81) and are resampled with
 * np.interp / interp1d before compute_LWIR_apparent_radiance (LWIR_HSI_Generator.py:151-167).
 *   Xk[nk] fp64 ascending knots; F[nk][nS] float32; out[nx][nS] float32; X == NULL -> grid.
 *   Outside [Xk[0], Xk[nk-1]] the end values are held, like np.interp. */
int rtx_interp_knots(const rtx_grid* grid, const double* X, int64_t nx, const double* Xk,
                     int64_t nk, const float* F, int64_t nS, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused at-sensor band radiances for MANY emissivity spectra given on knots (config C4):
 *   L[b][k] = ILS_b( tau*(eps_k*B(Ts) + (1-eps_k)*Ld) + La ),  eps_k = np.interp(X, Xk, E[:,k]).
 * = compute_LWIR_apparent_radiance (radiative_transfer.py:1064-1068, nA = 1, no dT) followed by
 * ILS_MAKO (:1236-1256 / ILS_MAKO.py:21-33), evaluated as (C_b + sum_j M[b][j] E[j][k]) / N_b:
 * rtx_band_moments makes ONE pass over the monochromatic tau/La/Ld (uniform grid) and writes
 *   N[nB], C[nB], M[nB][nk] float32 and jrange[nB][2] int32 (first/last knot each band touches);
 * rtx_band_mix contracts them with E[nk][nE] float32 -> out[nB][nE] float32.
 * Nothing of size nX*nE is ever formed (the reference needs a 1.4 TB temporary for this). */
int rtx_band_moments(int kind, const rtx_grid* grid, const float* tau, const float* La,
                     const float* Ld, double Ts, const double* Xk, int64_t nk, int nB,
                     const double* centre, const double* sigma, float* N_out, float* C_out,
                     float* M_out, int32_t* jrange_out, void* stream);
int rtx_band_mix(const float* N, const float* C, const float* M, const int32_t* jrange, int nB,
                 int64_t nk, const float* E, int64_t nE, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused HSI cube (config C5): every pixel has its own emissivity mixture and surface temperature
 * (LWIR_HSI_Generator.py:151-167: em = mixFrac . emis[ix_em], T = Ts + dT*N(0,1),
 * L = tau*(em*B(T) + (1-em)*Ld) + La), evaluated at monochromatic resolution and passed through
 * the ILS. B(nu, T_p) is interpolated over each band's support through Q Chebyshev nodes
 * (Q = 4: < 7e-9 relative over the MAKO bands, 230-350 K), which makes the monochromatic pass pixel-independent:
 *   rtx_band_basis_moments -> N[nB], C[nB], MLd[nB][nk], MB[Q][nB][nk], jrange[nB][2]
 *       basis_coef_h[Q][Q]: monomial coefficients of the Lagrange basis l_q(s), s = (nu-c_b)/(node_span*sigma_b)
 *       (MLd_out may be row Q of one [Q+1][nB][nk] array whose rows 0..Q-1 are MB_out)
 *   rtx_band_mix_stacked contracts the n_stack = Q+1 moment arrays M[n_stack][nB][nk] with the endmember knot
 *       spectra E[nk][nEnd] in one launch -> tab[nEnd][n_stack][nB] (bands innermost: what rtx_pixel_cube reads)
 *   rtx_pixel_cube -> cube[nB][nPix] float32 (bands first):
 *       tab[nEnd][Q+1][nB]: rows 0..Q-1 the Planck-node tables, row Q the downwelling table;
 *       kidx[nPix][nMix] int32 endmember indices, frac[nPix][nMix] float32, Tpix[nPix] fp64 (device);
 *       s_node_h[Q] the Chebyshev nodes in s. */
int rtx_band_basis_moments(int kind, const rtx_grid* grid, const float* tau, const float* La,
                           const float* Ld, const double* Xk, int64_t nk, int nB,
                           const double* centre, const double* sigma, int Q,
                           const float* basis_coef_h, double node_span, float* N_out,
                           float* C_out, float* MLd_out, float* MB_out, int32_t* jrange_out,
                           void* stream);
int rtx_band_mix_stacked(const float* M, const int32_t* jrange, int nB, int n_stack, int64_t nk,
                         const float* E, int64_t nE, float* tab, void* stream);
int rtx_pixel_cube(int nB, int Q, const double* centre, const double* sigma, double node_span,
                   const float* s_node_h, const float* N, const float* C, const float* tab,
                   int nEnd, int64_t nPix, int nMix, const int32_t* kidx, const float* frac,
                   const double* Tpix, float* cube, void* stream);

/* ------------------------------------------------------------------------------------------
 * Speed-dependent Voigt line-sum. Replaces the per-line PROFILE_SDVOIGT + scatter-add of
 * absorptionCoefficient_SDVoigt, misc/hapi.py:10897-10900 (PROFILE_SDVOIGT :10117 -> pcqsdhc :9850-10024 with
 * anuVC = eta = 0: PART1 for lines without speed dependence, PART2/3/4 otherwise; CPF = hum1_wei :9833, cpf3 :9645),
 * after rtx_line_prep_profile(..., RTX_PROFILE_SDVOIGT, ...). Evaluated in fp64 (the profile is a difference of two
 * complex probability functions): a workgroup per tile of 1024 points, far wings at Chebyshev nodes (32 per tile / 12 per
 * 64-point row, interpolation <= 1e-10 of a line's own contribution), the rows around a centre and around every regime
 * switch point by point; RADTXFR_SD_KERNEL=gather selects the one-thread-per-point cross-check. This is the
 * cross-section generator's path (misc/RT_gen_AbsXS_files.py:90), not the TUD hot path.
 *   out_f32[n_layers][ld] (may be NULL)  (float)(sum * scale);  out_f64[n_layers][ld] (may be NULL)  sum */
int rtx_sdvoigt_sum(const rtx_prep* prep, const rtx_grid* grid, int n_layers, float* out_f32,
                    double* out_f64, int64_t ld, void* stream);

/* ---- post-processing of TUD products: smooth / reduceResolution ---------------------
 * Replaces radiative_transfer.py:1266-1324 (smooth: reflect-padded window convolution) and :1327-1350
 * (reduceResolution: symmetrised smoothing + scipy cubic interp1d onto a coarser axis).
 *   rtx_fir_reflect: out[r][i] = sum_k taps_h[k] * in[r][R(i + k - centre)], i in [0,n), fp64 accumulation;
 *       R reflects about the first and last sample (j<0 -> -j, j>=n -> 2(n-1)-j), the reference's padding (:1314).
 *       in: [n_rows][ld_in] float32 (in_is_f64 = 0) or float64 (1), device; taps_h: n_taps doubles on the HOST
 *       (n_taps <= 8192, n_taps - 1 <= n); out: [n_rows][ld_out] float64, device.
 *   rtx_cubic_resample: the cubic spline through ALL samples of a uniform axis x0 + i*h, evaluated at x_out
 *       (device, fp64): in the interior scipy's not-a-knot interp1d(kind='cubic') is the cardinal cubic spline,
 *       whose coefficients are the samples filtered by sqrt(3)*(sqrt(3)-2)^|k|; truncated at |k| <= 40 (1e-23).
 *       Every x_out must lie in [x0 + 25 h, x0 + (n - 27) h] (error otherwise: the end conditions of the
 *       reference's spline, whose influence decays as 0.268^k, are not reproduced; at 25 knots it is 5e-15). Ysm: [n_rows][ld] float64; out: [n_rows][ld_out] float64. */
int rtx_fir_reflect(const void* in, int in_is_f64, int64_t ld_in, int n_rows, int64_t n,
                    const double* taps_h, int n_taps, int centre, double* out, int64_t ld_out,
                    void* stream);
int rtx_cubic_resample(const double* Ysm, int64_t ld, int n_rows, int64_t n, double x0, double h,
                       const double* x_out, int64_t n_out, double* out, int64_t ld_out, void* stream);
/* rtx_cubic_resample without its range check, which reads a flag back from the device and therefore synchronises the
 * stream: for a stream of spectra resampled onto ONE output axis that the caller has checked on the host (the loop of
 * Generate_LWIR_TUD.py:117-150 calls reduceResolution, radiative_transfer.py:1327-1350, once per atmosphere). Fully
 * asynchronous; an abscissa outside the supported range gives NaN. */
/* The end regions of the same spline (within a window length of either end of the axis the reference's knots are not
 * uniform -- it smooths the axis too, radiative_transfer.py:1331-1334, and smooth()'s reflection padding bends them -- and
 * the not-a-knot end condition acts): the not-a-knot spline on the m local samples Ysm[r][i_first .. i_first+m) with the TRUE
 * knots `knots[m]` (device, fp64; the smoothed axis there), natural at the cut; valid for x_out at least 24 knots inside
 * the cut (2e-14). high_end = 0: local sample 0 is the first sample of the axis; 1: local sample m-1 is the last one.
 * m <= 768. out: [n_rows][ld_out]. */
int rtx_cubic_end(const double* Ysm, int64_t ld, int n_rows, int64_t i_first, int m, int high_end,
                  const double* knots, const double* x_out, int64_t n_out, double* out, int64_t ld_out,
                  void* stream);
int rtx_cubic_resample_unchecked(const double* Ysm, int64_t ld, int n_rows, int64_t n, double x0, double h,
                                 const double* x_out, int64_t n_out, double* out, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single-process collectives over the GPUs of one node. The reference has no multi-GPU code; its scripts are plain
 * Python programs that fan work out with multiprocessing (Generate_LWIR_TUD.py:117-150). These three calls let ONE host
 * process shard a spectrum over several devices and reassemble it -- the one exchange of the path: the all-gather of the
 * packed [tau, L-up, L-down] blocks (radtxfr_amd/dist.py: compute_TUD_local) -- without a launcher.
 *   rtx_comm_init_all  devs_h[ndev] device indices. backend -1: RCCL (ncclCommInitAll, resolved at run time from the
 *                      librccl the process already holds, else librccl.so) when it loads and the devices are distinct,
 *                      otherwise peer copies; 0: peer copies (hipMemcpyPeerAsync fan-out; a device may repeat); 1: RCCL or
 *                      fail. Environment RADTXFR_COMM=peer|rccl overrides. Allocates and synchronises.
 *   rtx_comm_backend   0 peer copies, 1 RCCL
 *   rtx_allgather      rank i contributes sendbufs_h[i][0..count) float32 on device devs[i]; afterwards every
 *                      recvbufs_h[j][i*count + t] = sendbufs_h[i][t]. streams_h[i] is rank i's stream (send block complete
 *                      in its order before, gathered block complete in its order after). Asynchronous to the host. */
typedef struct rtx_comm rtx_comm;
int rtx_comm_init_all(int ndev, const int* devs_h, int backend, rtx_comm** out);
int rtx_comm_backend(const rtx_comm* comm);
int rtx_allgather(rtx_comm* comm, const void* const* sendbufs_h, void* const* recvbufs_h, int64_t count,
                  void* const* streams_h);
int rtx_comm_destroy(rtx_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* RADTXFR_HIP_H */
