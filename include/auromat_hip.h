/*
 * auromat_hip.h — C ABI of libauromat_hip.so
 *
 * MI355X (gfx950) implementation of the per-pixel georeferencing + resampling
 * hot path of esa/auromat.  The reference is pure Python with an import-time
 * two-backend switch (`_np` / `_ne`, e.g. auromat/coordinates/intersection.py:160-163);
 * this library is the third backend.  Each entry point names the reference
 * function(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - every function returns 0 on success, a negative AMT_E* code otherwise and
 *    never throws; amt_last_error() gives the message of the last failure on a context;
 *  - all array arguments are DEVICE pointers (hipMalloc'ed / torch .data_ptr()),
 *    C-contiguous, float64 unless stated; small fixed-size parameter blocks
 *    (3-vectors, 3x3 matrices, amt_frame_params) are HOST pointers;
 *  - kernels are enqueued on the context's stream and the call returns without
 *    synchronising unless documented ("synchronises");
 *  - the caller owns every buffer; the library keeps no global state and contexts
 *    may be used from different threads (one thread per context at a time);
 *  - missing data is NaN (reference convention: intersection.py:50-56), never an error.
 */
#ifndef AUROMAT_HIP_H
#define AUROMAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: amt_georef_out grew (bin_pole, altitude); amt_rotate_pole_deg, amt_pipe_finalize_stream, amt_seq_* added (round 2) */
/* 4: amt_georef_last_variant; the MLat / MLT-only mode of the fused frame kernel (see amt_georef_out); the box-first plan
 *    (amt_pipe_launch_box[_many], amt_pipe_launch_many_res, amt_plate_carree_resolution) (round 4) */
/* 5: the single-pass plan on caller-supplied corner directions (amt_georef_coarse_bbox_dirs, amt_pipe_coarse_dirs,
 *    amt_pipe_launch_dirs); griddata(method='cubic') exactly: amt_delaunay_*, amt_cubic_gradients_csr, amt_cubic_eval (round 5) */
/* 6: amt_georef_out.row_layout (strip-padded rows for buffers a pipeline owns), amt_padded_pitch, amt_unpad_rows; host images in the
 *    sequence runner (amt_run_frame.img_host, amt_run_result.uploaded_bytes, amt_georef_image_rows, amt_malloc_host / amt_free_host)
 *    amt_pipe_launch_dirs_many, amt_host_threads, AMT_EDOMAIN; retired: amt_linear_gather, amt_cubic_gradients, amt_cubic_gather
 *    (round 6) */
#define AMT_ABI_VERSION 6

#define AMT_OK 0
#define AMT_EINVAL (-1)   /* bad argument (NULL pointer, negative size, unsupported dtype ...) */
#define AMT_EHIP (-2)     /* a HIP runtime call failed; see amt_last_error */
#define AMT_ENOMEM (-3)
#define AMT_EEMPTY (-4)   /* operation would leave no valid pixel (mapping.py:858-859 -> ValueError) */
#define AMT_EDOMAIN (-5)  /* the result exists but nothing can be built on it (amt_plate_carree_resolution: no longitude resolution
                           * for a box that goes all the way round; ABI v6) */

typedef struct amt_ctx amt_ctx;

/* ---- context & memory --------------------------------------------------------------- */

int amt_abi_version(void);
/* stream: the hipStream_t to enqueue on, used as given — NULL is the device's default (null)
   stream, which is also what torch.cuda.current_stream().cuda_stream reports unless a side stream
   is active.  own_stream != 0: ignore `stream`, create and own a non-blocking stream instead. */
int amt_ctx_create(int device_id, void* stream, int own_stream, amt_ctx** out_ctx);
int amt_ctx_destroy(amt_ctx* ctx);
/* A context may be moved between streams (the Python host follows torch's current stream): every call enqueues on
   the stream set at that moment, and the library's internal scratch memory is kept per stream, so work enqueued
   on different streams through one context may overlap on the GPU.  The context itself is not thread-safe: one
   host thread at a time. */
int amt_ctx_set_stream(amt_ctx* ctx, void* stream);
void* amt_ctx_get_stream(amt_ctx* ctx);
int amt_ctx_synchronize(amt_ctx* ctx);                 /* synchronises */
const char* amt_last_error(amt_ctx* ctx);
/* Device properties the host side prints next to roofline numbers. */
int amt_device_info(amt_ctx* ctx, char* name, size_t name_len, int* compute_units, int* clock_khz,
                    size_t* total_mem);

/* Plain device-memory helpers so that a host without torch can drive the library. */
int amt_malloc(amt_ctx* ctx, size_t bytes, void** out_dptr);
int amt_free(amt_ctx* ctx, void* dptr);
/* Host threads a pool of this process may use when it would like `wanted` (no GPU call; ABI v6): the cores the process may run on
 * (sched_getaffinity) divided by the ranks that share the node (LOCAL_WORLD_SIZE of torch.distributed.run, or AMT_LOCAL_RANKS),
 * at least 1, at most `wanted`.  The library's own pools (amt_upload_staged's copy threads, the triangulator's strips) are
 * sized by it, so that eight ranks on one host do not start 8 x (8 + 16) threads on its cores.  *cores / *local_ranks (each
 * optional) report what it saw.  The reference is single-threaded per process (mapping/spacecraft.py:326-332). */
int amt_host_threads(int wanted, int* cores, int* local_ranks);
/* Page-locked host memory (what amt_run_frame.img_host points into: the DMA engine reads it at the link's rate while the host
 * goes on); ABI v6. */
int amt_malloc_host(amt_ctx* ctx, size_t bytes, void** out_hptr);
int amt_free_host(amt_ctx* ctx, void* hptr);
int amt_memcpy_h2d(amt_ctx* ctx, void* dst, const void* src, size_t bytes);   /* async on the stream */
int amt_memcpy_d2h(amt_ctx* ctx, void* dst, const void* src, size_t bytes);   /* synchronises */
int amt_memset(amt_ctx* ctx, void* dst, int value, size_t bytes);
/* Whole arrays between PAGEABLE host memory and the device at the rate of the link: a few host threads (AMT_COPY_THREADS,
 * default 8) each copy every n-th 4 MiB piece through page-locked staging pieces of their own and hand it to the DMA engine
 * on the context's stream (what a NumPy-array API like the reference's needs at its boundary: mapping.py:318-337 takes the
 * image as an array, every property returns one).  amt_upload_staged returns when the last piece has been handed over (the
 * source may be reused; the device side is ordered on the stream); amt_download_staged returns when dst_host is complete. */
int amt_upload_staged(amt_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int amt_download_staged(amt_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
/* HIP-event timing on the context's stream (bench.py measures kernels with these). */
int amt_event_create(amt_ctx* ctx, void** out_event);
int amt_event_destroy(amt_ctx* ctx, void* event);
int amt_event_record(amt_ctx* ctx, void* event);
int amt_event_elapsed_ms(amt_ctx* ctx, void* start, void* stop, float* out_ms);   /* synchronises on stop */
/* Per-kernel timing: while enabled, amt_georef_frame[_dirs] and amt_bin_frame bracket their main kernel
 * launch (k_georef_rows / k_bin_frame, not the small fold / finalize kernels) with HIP events on the
 * context's stream.  enable = n > 0 brackets every n-th launch of each kind (the two event packets sit between
 * consecutive kernels on the stream, so a pipelined caller samples instead of timing every launch); 0 = off.
 * amt_timing_read sums the recorded launches and returns how many frames they covered (synchronises; a launch of
 * amt_pipe_launch_many covers several frames, so total / frames is the time per frame); enabling resets. */
#define AMT_KERNEL_GEOREF 0
#define AMT_KERNEL_BIN 1
int amt_timing_enable(amt_ctx* ctx, int enable);
int amt_timing_read(amt_ctx* ctx, int kernel, double* total_ms, int* launches);
/* Which variant of the frame kernel the context's latest amt_georef_frame[_dirs] / amt_pipe_launch[_many] / amt_run_* launch
 * ran (diagnostics; tests assert on it): *second = 0 (lat, lon) only, 1 + (MLat, MLT), 2 / 3 the pole plans, 4 the
 * MLat / MLT-only mode (see amt_georef_out); *bin = 0 no fused binning, 1 uint8, 2 uint16 image; *frames = frames in that
 * launch.  -1 in all three before the first launch. */
int amt_georef_last_variant(amt_ctx* ctx, int32_t* second, int32_t* bin, int32_t* frames);

/* ---- per-frame parameter block --------------------------------------------------------
 * Host scalars the reference derives once per frame:
 *   cd, crpix, rot   auromat/coordinates/wcs.py:80-99,135-139 (TAN WCS; rot = euler_matrix(...,'rzxz')[:3,:3])
 *   cam              BaseMapping.cameraPosGCRS, auromat/mapping/mapping.py:318-337
 *   a, b             wgs84A/B + altitude, auromat/mapping/mapping.py:1498-1501
 *   a0, b0           wgs84A/B for ECEF->geodetic, auromat/coordinates/transform.py:338
 *   m_geo, m_sm      mat_j2000_to_geo / mat_j2000_to_sm, auromat/coordinates/transform.py:683-691
 */
typedef struct amt_frame_params {
    int32_t width;        /* IMAGEW */
    int32_t height;       /* IMAGEH */
    int32_t fast_center;  /* BaseAstrometryMapping.fastCenterCalculation, astrometry.py:23-40 */
    int32_t reserved;
    double cd[4];         /* CD1_1 CD1_2 CD2_1 CD2_2 [deg/px] */
    double crpix[2];      /* CRPIX1 CRPIX2 (1-based FITS convention) */
    double rot[9];        /* native -> celestial rotation, row major */
    double cam[3];        /* camera position, km, J2000/GCRS */
    double a, b;          /* inflated ellipsoid axes, km */
    double a0, b0;        /* geodetic reference ellipsoid axes, km */
    double m_geo[9];      /* J2000 -> GEO, row major */
    double m_sm[9];       /* J2000 -> SM, row major */
} amt_frame_params;

/* Output block of amt_georef_frame.  Any pointer may be NULL (that array is not written).
 * Corner arrays have (height+1)*(width+1) elements, centre arrays height*width.
 * bbox (optional, 8 doubles, device): [lat_min, lat_max, lon_min, lon_max, lon_min_positive,
 * lon_max_nonpositive, n_valid_centres, 0] over the corners of centres with
 * elevation >= bbox_min_elevation — the inputs of BaseMapping.boundingBox
 * (auromat/mapping/mapping.py:693-743) for a mapping that was maskedByElevation()'d
 * (mapping.py:845-864).  Slot 7 (pole containment, what geodesic.py:183 containsOrCrossesPole decides
 * from the outline) is left 0 here: with a camera model the host projects the pole into the frame
 * instead; amt_bbox_corners fills it for arbitrary grids. */
struct amt_axis;                    /* defined in the binning section below */

typedef struct amt_georef_out {
    double* lat;      /* corners, deg   (BaseAstrometryMapping.lats,  astrometry.py:118-144) */
    double* lon;      /* corners, deg */
    double* lat_c;    /* centres, deg   (latsCenter / lonsCenter, astrometry.py:128-152) */
    double* lon_c;
    double* elev;     /* centres, deg   (elevation, astrometry.py:200-212) */
    double* mlat;     /* corners, deg   (mLatMlt, astrometry.py:170-183) */
    double* mlt;      /* corners, hours */
    double* mlat_c;   /* centres        (mLatMltCenter, astrometry.py:185-198) */
    double* mlt_c;
    double* bbox;
    double bbox_min_elevation;
    /* Optional fused binning (single-pass resample(method='mean'), auromat/resample.py:301-351): when
     * bin_acc != NULL every pixel with elevation >= bbox_min_elevation is binned right where it is
     * computed — x = lon_c, y = lat_c, or (SM longitude, MLat) with bin_magnetic, or the pole-rotated
     * (lon, lat) with bin_pole — into the uint64
     * accumulator planes of amt_bin_frame (count, 3 channel sums, fixed-point elevation), so the centre
     * arrays need not be read back (and need not be written at all: lat_c/lon_c/elev may be NULL).
     * The grid must be known before the launch: callers use a superset of the final grid, aligned to the
     * same global nodes (amt_georef_coarse_bbox), and crop in amt_bin_frame_finalize_window.
     * Both axes must be uniform; bin_img is (height, width, 3) uint8 (dtype 1) or uint16 (dtype 2).
     * MLat / MLT-only mode: with bin_magnetic, no bin_pole and lat = lon = lat_c = lon_c = NULL the kernel computes what
     * resampleMLatMLT consumes and nothing else (reference resample.py:63-71, mapping.py:1519-1547: mLatMlt, mLatMltCenter,
     * elevation, image) — ray, shell, SM rotation, (MLat, SM longitude), elevation, bin; no ECEF -> geodetic step at all.
     * mlat / mlt / mlat_c / mlt_c / elev (each optional) and the grids are bit-identical to the nine-array mode. */
    const struct amt_axis* bin_xaxis;     /* host pointers */
    const struct amt_axis* bin_yaxis;
    const void* bin_img;
    uint64_t* bin_acc;
    int32_t bin_img_dtype;
    int32_t bin_lon_wrap;
    int32_t bin_magnetic;          /* != 0: x = SM longitude (mltToSmLon(mlt)), y = MLat; bbox[0..6] then refer to these too.
                                    * Without bin_acc: only that — the box in (MLat, SM longitude); lat, lon, lat_c, lon_c
                                    * must then be NULL (MLat / MLT-only mode without binning) */
    /* Scheduling hint, no effect on results: order in which the frame's work items (strips of 63 columns x 16
     * rows, row-major) are dispatched.  1 = rows top to bottom, 2 = bottom to top, 3 = interleaved (rows of items in
     * the order (k * s) mod n with s near n / golden ratio; measured 7 % slower than 1 / 2 for a kernel run alone, equal
     * inside the pipeline); 0 = automatic: bottom to top when the nadir lies below the frame centre (camera model;
     * top to bottom for caller-supplied directions).  Rays that miss the shell are cheap, hits are expensive;
     * starting with the rows where the Earth is lets the cheap items fill the end of the launch (4-5 % shorter
     * kernel; when the Earth is to the left or right every row mixes both kinds anyway).  amt_georef_coarse_bbox
     * reports the side from actual hits (bbox[7]). */
    int32_t item_order;
    /* Optional, with bin_acc: pixels that sit ON a bin edge in the sense of the right-most-edge rule
     * (histogram.py:215-224) are then not binned but appended to `bin_events` (32-byte records, see
     * csrc/amt_common.h bin_event; device memory, room for bin_event_capacity records) and counted in
     * *bin_event_count (device uint32, zero before the launch; it keeps counting beyond the capacity, which tells
     * the caller that the frame must be redone).  Whether such a pixel belongs to the bin above or below its edge
     * depends on whether the edge is the last one of the FINAL grid; the frame driver resolves them in
     * amt_pipe_finalize.  NULL: they are binned above the edge, exact only when the grid given is the final one. */
    void* bin_events;
    uint32_t* bin_event_count;
    int64_t bin_event_capacity;
    /* Pole plan of the fused binning (a pole of the grid's coordinates is in view; not with bin_lon_wrap): pixels are
     * binned at (lat, lon) — with bin_magnetic at (MLat, SM longitude) taken as if they were geodetic, as
     * resampleMLatMLT does — rotated by +90 deg about x at `altitude`: amt_rotate_pole_deg of the centre coordinates,
     * reference resample.py:176-201 — and bbox[0..5] are reduced over the rotated corners (to ~1e-11 deg: good for laying out the
     * grid unless an extreme sits within that of a grid node, see amt_pipe_wait).  `altitude` [km] is the mapping
     * altitude the shell (a, b) = (a0, b0) + altitude was built from, as rotatePole takes it.  MLat / MLT outputs can be
     * combined with either grid. */
    int32_t bin_pole;
    /* Row layout of the nine per-pixel output arrays (ABI v6; 0 in earlier callers' zero-initialised structs):
     *   AMT_ROWS_CONTIGUOUS (0)    C-contiguous (height+1, width+1) / (height, width), what the reference's properties return
     *                              (astrometry.py:118-152);
     *   AMT_ROWS_STRIP_PADDED (1)  for buffers a pipeline OWNS and hands on only on request: every array has rows of
     *                              P = amt_padded_pitch(width) doubles (corner arrays height+1 rows, pixel arrays height rows) and
     *                              element (y, x) lies at y P + 64 (x / 63) + x % 63 — the kernel's work items are strips of 63
     *                              columns, and with this layout each strip's run of a row is a whole number of 128-byte lines
     *                              that no other wave writes (a measured 8-12 % of the kernel's time: DESIGN.md 4.1).  The 64th
     *                              double of a strip (and columns beyond the frame in the last strip) is padding with
     *                              unspecified content.  amt_unpad_rows compacts an array into contiguous rows, bit for bit the
     *                              array the contiguous layout gives.  Row-marching kernel only (not AMT_GEOREF_KERNEL=tile);
     *                              the two-pass binning (amt_bin_frame) and every other consumer read contiguous rows. */
    int32_t row_layout;
    double altitude;
} amt_georef_out;
#define AMT_ROWS_CONTIGUOUS 0
#define AMT_ROWS_STRIP_PADDED 1
/* Doubles per row of a strip-padded array of a frame `width` pixels wide (64 per strip of 63 columns; 4352 for 4240). */
int64_t amt_padded_pitch(int32_t width);
/* Strip-padded rows -> contiguous rows (device to device, on the context's stream): `src` has `rows` rows of
 * amt_padded_pitch(width) doubles, `dst` becomes a C-contiguous (rows, cols) array with cols = width (pixel arrays) or
 * width + 1 (corner arrays) — what BaseAstrometryMapping.lats & co. hand out (astrometry.py:118-152). */
int amt_unpad_rows(amt_ctx* ctx, const double* src_padded, int32_t rows, int32_t cols, int32_t width, double* dst);

/* ---- building blocks (auromat.coordinates) ------------------------------------------- */

/* auromat/coordinates/wcs.py:18-64,66-144 pix2world(..., ascartesian=True) for TAN headers and
 * auromat/mapping/astrometry.py:245-269 pixelDirection: unit direction of every pixel corner
 * (corner=1: (height+1, width+1, 3)) or centre (corner=0: (height, width, 3)), AoS. */
int amt_directions_tan(amt_ctx* ctx, const amt_frame_params* p, int corner, double* out_dirs);
/* auromat/coordinates/wcs.py:54-56: headers that are not plain TAN go to astropy.wcs.WCS(header).all_pix2world in the reference.
 * The zenithal family (Calabretta & Greisen 2002, 5.1: TAN, SIN without slant, ARC, STG, ZEA) with SIP distortion polynomials
 * (CTYPE "...-SIP": A_p_q / B_p_q, Shupe et al. 2005) on the device: unit direction (J2000) of every pixel corner (corner = 1:
 * (height+1, width+1, 3)) or centre of the rectangle that starts at pixel (start_x, start_y).  rot: native -> celestial
 * rotation, euler_matrix(RA+90, 90-Dec, -(LONPOLE-90), 'rzxz') as for amt_frame_params.  sip_a[p][q] multiplies u^p v^q
 * (u, v: pixel offsets from CRPIX), zero beyond the order; order 0 = no distortion.  The result feeds amt_georef_frame_dirs. */
#define AMT_SIP_MAX 10
typedef struct amt_zenithal_wcs {
    int32_t width, height, corner, projection;      /* projection: 0 TAN, 1 SIN, 2 ARC, 3 STG, 4 ZEA */
    double cd[4], crpix[2], rot[9];
    double start_x, start_y;
    int32_t sip_order_a, sip_order_b;
    double sip_a[AMT_SIP_MAX][AMT_SIP_MAX], sip_b[AMT_SIP_MAX][AMT_SIP_MAX];
} amt_zenithal_wcs;
int amt_directions_zenithal(amt_ctx* ctx, const amt_zenithal_wcs* w, double* out_dirs);
/* auromat/coordinates/wcs.py:66-144 tan_pix2world(header, px, py, origin, ascartesian=True) for arbitrary
 * pixel coordinates (origin 0 or 1 as in FITS/astropy); out (n,3) AoS.  Uses cd, crpix, rot of p only. */
int amt_directions_tan_points(amt_ctx* ctx, const amt_frame_params* p, const double* px, const double* py,
                              int64_t n, int origin, double* out_dirs);

/* auromat/coordinates/intersection.py:144-163 ellipsoidLineIntersection (and
 * auromat/mapping/mapping.py:1474-1510 inflatedEarthIntersection with a=wgs84A+h, b=wgs84B+h).
 * origin: host double[3]; dirs/out: (n,3) AoS.  Misses / points behind a directed ray are NaN. */
int amt_intersect_ellipsoid(amt_ctx* ctx, double a, double b, const double* origin, const double* dirs,
                            int64_t n, int directed, double* out_xyz);
/* auromat/coordinates/intersection.py:229-237 ellipsoidLineIntersects -> uint8 (0/1). */
int amt_intersects_ellipsoid(amt_ctx* ctx, double a, double b, const double* origin, const double* dirs,
                             int64_t n, int directed, uint8_t* out_hit);
/* auromat/coordinates/intersection.py:12-48 sphereLineIntersection (earthModel='sphere'). */
int amt_intersect_sphere(amt_ctx* ctx, double radius, const double* origin, const double* dirs,
                         int64_t n, int directed, double* out_xyz);

/* auromat/coordinates/transform.py:199-297 ecef2Geodetic (Bowring 1985) -> radians. */
int amt_ecef_to_geodetic(amt_ctx* ctx, const double* x, const double* y, const double* z, int64_t n,
                         double a, double b, double* out_lat, double* out_lon);
/* auromat/coordinates/transform.py:156-178 geodetic2Ecef (radians in, scalar height). */
int amt_geodetic_to_ecef(amt_ctx* ctx, const double* lat, const double* lon, double h, int64_t n,
                         double a, double b, double* out_x, double* out_y, double* out_z);
/* auromat/coordinates/transform.py:324-343 j2000ToLatLon with the 3x3 (host, row major) given:
 * rotate (n,3) AoS points, Bowring, degrees out.  Also serves x_to_y-based geo helpers. */
int amt_rotate_to_latlon(amt_ctx* ctx, const double* m, const double* xyz, int64_t n, double a0, double b0,
                         double* out_lat_deg, double* out_lon_deg);
/* auromat/coordinates/transform.py:403-459 j2000ToMLatMLT / geoToMLatMLT: rotate into SM,
 * mlat = deg(atan2(z, hypot(x,y))), mlt = deg(atan2(y,x))*24/360 + 12. */
int amt_rotate_to_mlat_mlt(amt_ctx* ctx, const double* m, const double* xyz, int64_t n,
                           double* out_mlat_deg, double* out_mlt_h);
/* auromat/coordinates/transform.py:728-738 x_to_y: out = m @ v for (n,3) AoS vectors. */
int amt_rotate_vectors(amt_ctx* ctx, const double* m, const double* xyz, int64_t n, double* out_xyz);
/* auromat/mapping/mapping.py:540-550 BaseMapping._mLatMlt: geodetic (deg) at height h -> ECEF ->
 * SM (m = mat_geo_to_sm) -> MLat/MLT.  NaN in, NaN out. */
int amt_latlon_to_mlat_mlt(amt_ctx* ctx, const double* m, const double* lat_deg, const double* lon_deg,
                           double h, int64_t n, double a0, double b0, double* out_mlat_deg, double* out_mlt_h);
/* auromat/coordinates/transform.py:461-485 smToLatLon: SM lat/lon (deg) on the unit sphere ->
 * GEO (m = transpose of mat_geo_to_sm, passed already transposed) -> geodetic deg. */
int amt_sm_to_latlon(amt_ctx* ctx, const double* m_sm_to_geo, const double* smlat_deg, const double* smlon_deg,
                     int64_t n, double a0, double b0, double* out_lat_deg, double* out_lon_deg);
/* auromat/coordinates/transform.py:301-322 rotatePole: geodetic (rad) at `altitude` rotated by the
 * 3x3 `rot` (host) about the origin, back to geodetic (rad). */
int amt_rotate_pole(amt_ctx* ctx, const double* rot, const double* lat, const double* lon, double altitude,
                    int64_t n, double a0, double b0, double* out_lat, double* out_lon);
/* The same with degrees in and out (lat_deg * pi/180 -> amt_rotate_pole -> * 180/pi, rounding for rounding what
 * np.deg2rad / np.rad2deg around the call give): what the pole branch of the resampling works in
 * (auromat/resample.py:176-201,262-273). */
int amt_rotate_pole_deg(amt_ctx* ctx, const double* rot, const double* lat_deg, const double* lon_deg, double altitude,
                        int64_t n, double a0, double b0, double* out_lat_deg, double* out_lon_deg);
/* auromat/coordinates/transform.py:142-154 cartesian_to_spherical -> (r, lat, lon) radians; out_r may be NULL. */
int amt_cartesian_to_spherical(amt_ctx* ctx, const double* x, const double* y, const double* z, int64_t n,
                               double* out_r, double* out_lat, double* out_lon);
/* auromat/coordinates/transform.py:89-102 spherical_to_cartesian; r may be NULL (unit sphere). */
int amt_spherical_to_cartesian(amt_ctx* ctx, const double* r, const double* lat, const double* lon, int64_t n,
                               double* out_x, double* out_y, double* out_z);

/* ---- other camera models feeding the same intersection + geodetic steps -------- */

/* Equidistant all-sky (fisheye) calibration of one ground station, auromat/mapping/miracle.py:28-35,314-347
 * (FMI MIRACLE cal.txt: zenith pixel (xc vertical, yc horizontal), d = k*z, CCW rotation in rad), already
 * scaled to the image size (miracle.py:320-326), with the station's position and local-frame rotation. */
typedef struct amt_allsky_params {
    int32_t size;             /* image is size x size pixels */
    int32_t reserved;
    double xc, yc, k;         /* pixels, pixels, pixels per radian of zenith angle */
    double rotation;          /* radians */
    double center_offset;     /* added to the integer index of a pixel CENTRE (0.5; the reference run under its
                                 pinned NumPy 1.6 adds 0: `ind += 0.5` on an integer array, miracle.py:333-334) */
    double to_geo[9];         /* row-major R_z(-lon) . R_y(90deg - lat), miracle.py:249-252 */
    double station[3];        /* geodetic2EcefZero(lat, lon) in km, miracle.py:139-140 */
    double a, b;              /* shell semi-axes in km: wgs84A + altitude, wgs84B + altitude */
    double a0, b0;            /* wgs84A, wgs84B */
} amt_allsky_params;

/* MIRACLEMapping.calculateAzEl -> _calculateCameraToPixelDirection -> intersectionInflated* -> _calculateLatsLons
 * (auromat/mapping/miracle.py:196-258,314-347) for every pixel corner (corner=1: (size+1)^2 points) or centre
 * (corner=0: size^2 points) in one launch.  Every output may be NULL: az_deg in [0,360), el_deg = 90 - deg(d/k),
 * dirs (n,3) AoS unit vectors in GEO, lat_deg / lon_deg of the shell intersection. */
int amt_georef_allsky(amt_ctx* ctx, const amt_allsky_params* p, int corner, double* az_deg, double* el_deg,
                      double* dirs, double* lat_deg, double* lon_deg);

/* auromat/mapping/themis.py:224-253 reproject: coordinates given on the shell at height_ref (km above the
 * ellipsoid) as seen from the station at (station_lat_deg, station_lon_deg, height 0) -> the same lines of sight
 * on the shell at height_new.  Degrees in and out; NaN in, NaN out. */
int amt_reproject_altitude(amt_ctx* ctx, double station_lat_deg, double station_lon_deg, const double* lat_ref_deg,
                           const double* lon_ref_deg, int64_t n, double height_ref, double height_new,
                           double a0, double b0, double* out_lat_deg, double* out_lon_deg);

/* ---- fused frame kernel ------------------------------------------------------------- */

/* All lazy arrays of a BaseAstrometryMapping in one launch
 * (auromat/mapping/astrometry.py:49-212: cameraToPixel*Direction -> intersectionInflated* ->
 * _latsLonsCorner/_latsLonsCenter -> elevation -> mLatMlt[Center]).  Directions are generated
 * in-kernel from the WCS; fast_center selects the corner-mean centres (astrometry.py:100-101,154-160)
 * or the exact per-centre ray cast (:103-105). */
int amt_georef_frame(amt_ctx* ctx, const amt_frame_params* p, const amt_georef_out* out);
/* Same with caller-supplied corner directions ((height+1, width+1, 3) AoS, e.g. from another
 * camera model; SURVEY.md §8d "directions-in" variant).  fast_center must be 1. */
int amt_georef_frame_dirs(amt_ctx* ctx, const amt_frame_params* p, const double* corner_dirs,
                          const amt_georef_out* out);
/* Cheap estimate of the same bbox[0..6] from every `stride`-th pixel corner in both directions (a
 * 1/stride^2 sample of the rays): a corner counts when the elevation of its own ray is >= min_elevation.
 * With magnetic == 1 the box is in (MLat, SM longitude) instead of (lat, lon), with magnetic == 2 in (lat, lon)
 * rotated by +90 deg about x at the altitude a - a0 (the pole plan, amt_georef_out.bin_pole), with magnetic == 3 in
 * (MLat, SM longitude) rotated likewise.  Used to lay out a superset
 * grid for the fused binning before the full kernel runs; the caller adds a safety margin and checks the
 * exact box afterwards.  bbox[7] = sx * 2^20 + sy, where sx (sy) is the number of sampled rays that hit the
 * shell right of (below) the frame centre minus those left of (above) it: the input of amt_georef_out.item_order. */
int amt_georef_coarse_bbox(amt_ctx* ctx, const amt_frame_params* p, int32_t stride, double min_elevation,
                           int magnetic, double* bbox);
/* The same estimate for caller-supplied corner directions (amt_georef_frame_dirs): every `stride`-th direction of
 * corner_dirs ((height+1, width+1, 3), J2000) is cast instead of the TAN model's.  magnetic 0 or 1 (direction arrays have
 * no pole plan). */
int amt_georef_coarse_bbox_dirs(amt_ctx* ctx, const amt_frame_params* p, const double* corner_dirs, int32_t stride,
                                double min_elevation, int magnetic, double* bbox);
/* Host function (no GPU call): which rows of amt_georef_frame's work items cannot see the shell.  With the TAN camera
 * model the limb is a conic section in the image and the set of pixel corners whose ray hits the shell is convex, so
 * whole bands of the frame are bounded exactly from a handful of evaluations; the waves of such bands write NaN and
 * cast no ray.  An item is `rows_per_item` pixel rows tall; the frame has `n_item_rows` rows of items; rows
 * [0, top_end) and [bottom_begin, n_item_rows) are free of hits (conservative: within 1e-9 of the limb a band counts
 * as seeing the Earth).  Replaces nothing in the reference — it evaluates every ray (wcs.py:18-144,
 * intersection.py:58-104) — and changes no result. */
int amt_georef_sky_rows(const amt_frame_params* p, int32_t* rows_per_item, int32_t* n_item_rows, int32_t* top_end,
                        int32_t* bottom_begin);
/* Host function (no GPU call): the pixel rows [*row_begin, *row_end) of a camera frame's IMAGE that the fused binning can need
 * — a pixel's colours are read only to bin it, and only pixels whose ray hits the shell with an elevation >= min_elevation are
 * binned (reference mapping.py:845-864 maskedByElevation, resample.py:119-120,315-321).  Bounded from the camera model alone: the
 * limb as above, and the cone about the nadir inside which the elevation can reach min_elevation (law of sines in the triangle
 * Earth's centre / camera / hit point: sin(nadir angle) = |P| / |C| cos(elevation), |P| <= the shell's larger semi-axis).
 * Conservative, at the granularity of the work items' rows; min_elevation <= 0 or -inf: the rows a ray can hit.  What a host
 * that holds the image in page-locked memory has to send (amt_run_frame.img_host does exactly that). */
int amt_georef_image_rows(const amt_frame_params* p, double min_elevation, int32_t* row_begin, int32_t* row_end);

/* ---- mask rules ---------------------------------------------------------------------- */

/* auromat/mapping/mapping.py:845-864 maskedByElevation + the lazy _doSanitize(afterMasking=True)
 * of mapping.py:1063-1125,1161-1213.  Masks are uint8, 1 = masked.
 * center_mask[h*w] = !(elev >= min_elevation) (NaN counts as masked);
 * corner_mask[(h+1)*(w+1)] = isnan(corner_lat) | all adjacent centres masked.
 * n_valid (device int64, optional) receives the number of unmasked centres; the host raises
 * ValueError when it is 0 (AMT_EEMPTY is not raised here because the call does not synchronise). */
int amt_mask_by_elevation(amt_ctx* ctx, const double* elev, const double* corner_lat, int32_t height,
                          int32_t width, double min_elevation, uint8_t* center_mask, uint8_t* corner_mask,
                          int64_t* n_valid);
/* auromat/mapping/mapping.py:1063-1125 _doSanitize on boolean masks (in place).
 * img_mask may be NULL; after_masking as in the reference. */
int amt_sanitize_masks(amt_ctx* ctx, uint8_t* corner_mask, uint8_t* center_mask, const uint8_t* img_mask,
                       int32_t height, int32_t width, int after_masking);
/* auromat/utils.py:97-151 outline (skimage.measure.find_contours(padded mask, 0.99), rounded): the contour links of
 * a (height, width) uint8 mask (1 = masked).  For every unmasked pixel p (flat index i) and every direction d
 * (0 up, 1 right, 2 down, 3 left) whose 4-neighbour is masked or outside, one record links[2k] = 4*i + d,
 * links[2k+1] = the key of the crossing that follows it on the contour (unmasked region on the right-hand side, i.e.
 * clockwise in image coordinates; unmasked pixels are joined through edges only).  Records come in no particular order;
 * *count (device uint64, zeroed by the call) receives their number, which may exceed `capacity` (then only the first
 * `capacity` were written: call again with a larger buffer).  Following the links from any key yields one closed
 * contour; the pixel of each key (key / 4), with consecutive duplicates dropped, is the polygon the reference traces. */
int amt_mask_outline_links(amt_ctx* ctx, const uint8_t* mask, int32_t height, int32_t width, int64_t* links,
                           int64_t capacity, uint64_t* count);
/* auromat/draw_helpers.py:34-94 createPolygonsAndColors + filterNanPolygons for the pixels listed in `index` (n flat
 * pixel indices, normally the unmasked ones in row-major order): verts (n,4,2) float64 = (lat, lon) of the corners
 * (r,c), (r,c+1), (r+1,c+1), (r+1,c); colours per pixel as the mapping's `rgb` gives them (mapping.py:980-1007: uint8 as
 * is, uint16 * (255/65535) truncated; one channel is repeated): colors_u8 (n,3) and / or colors_f64 (n,3) = rgb / 255
 * (ColorMode.matplotlib); either may be NULL.  lat/lon: (height+1, width+1); img: (height, width, nchan), nchan 1 or 3. */
int amt_pixel_polygons(amt_ctx* ctx, const double* lat, const double* lon, const void* img, int32_t img_dtype,
                       int32_t nchan, int32_t height, int32_t width, const int64_t* index, int64_t n, double* verts,
                       uint8_t* colors_u8, double* colors_f64);
/* Inputs of BaseMapping.boundingBox (mapping.py:693-743) for arbitrary corner grids:
 * bbox[0..5] = min/max over unmasked corners as in amt_georef_out.bbox, bbox[6] = number of unmasked
 * corners, bbox[7] = number of unmasked centres whose corner quad winds around a pole.
 * corner_mask / center_mask may be NULL (then NaN latitude = masked; a centre is unmasked when its
 * four corners are). lat/lon: (height+1, width+1) degrees. */
int amt_bbox_corners(amt_ctx* ctx, const double* lat, const double* lon, const uint8_t* corner_mask,
                     const uint8_t* center_mask, int32_t height, int32_t width, double* bbox);

/* ---- histogram binning / plate-carree resampling ------------------------------------- */

/* Bin-edge description of one axis, reference semantics of auromat/util/histogram.py:178-224:
 * index = searchsorted(edges, v, 'right') (0 and nbin+1 are outliers), and values v >= edges[nbin]
 * with rint(v*scale)/scale == last_rounded fall into the last bin.  `scale` = 10**decimal and
 * `last_rounded` = around(edges[-1], decimal) are computed by the host exactly as the reference does.
 * uniform = 0: `edges` is a DEVICE array of nbin+1 ascending doubles, searched by bisection.
 * uniform = 1: the edges are exactly what np.linspace(first, last, nbin+1) produces, i.e.
 *   edges[i] = fl(fl(i*step) + first) for i < nbin (two roundings, no FMA), edges[nbin] = last, with
 *   step = (last-first)/nbin; the kernel evaluates them in registers (O(1) guess + exact fix-up) and
 *   `edges` may be NULL.  The host must have verified that identity for its edge array. */
typedef struct amt_axis {
    const double* edges;
    int32_t nbin;
    int32_t uniform;
    double first, last;   /* edges[0] and edges[nbin] */
    double step;          /* uniform axes: (last-first)/nbin */
    double scale;
    double last_rounded;
} amt_axis;

/* auromat/util/histogram.py:284-417,57-282 histogram2d(x, y, bins, range, weights=[None, w0, ...]).
 * Accumulates into count[nx*ny] and sums[k][nx*ny] (k < nweights; device pointers in the HOST array
 * `weights` / `sums`), all float64, row major (ix*ny + iy), which must be zeroed by the caller
 * (they may be accumulated over several calls).  NaN x or y are outliers; NaN weights poison
 * their cell (np.bincount behaviour).  lon_wrap != 0 bins wrap_at(x + 180, 180) instead of x
 * (auromat/resample.py:212-218, discontinuity branch). */
int amt_hist2d_accumulate(amt_ctx* ctx, const double* x, const double* y, int64_t n,
                          const double* const* weights, int32_t nweights, const amt_axis* xaxis,
                          const amt_axis* yaxis, int lon_wrap, double* count, double* const* sums);

/* Fused binning of one frame for resample(method='mean') (auromat/resample.py:95-142,301-351):
 * pixels with finite lat_c, elev >= min_elevation (pass -inf to disable; NaN elev is dropped when a
 * finite threshold is given) and center_mask == 0 (mask may be NULL) add 1, their image channels
 * and their elevation to cell (ix, iy), x = lon_c, y = lat_c.
 * img: (n, nchan) interleaved uint8 (img_dtype=1) or uint16 (img_dtype=2), nchan <= 4.
 * acc: device uint64[(nchan+2) * nx*ny] workspace, zeroed by the caller:
 *   plane 0 = count, planes 1..nchan = exact integer channel sums,
 *   plane nchan+1 = elevation sum in signed 31.32 fixed point (|error| <= 2^-33 deg per sample;
 *   order independent, so results are bit-reproducible). */
int amt_bin_frame(amt_ctx* ctx, const double* lat_c, const double* lon_c, const double* elev,
                  const void* img, int32_t img_dtype, int32_t nchan, const uint8_t* center_mask,
                  int32_t height, int32_t width, double min_elevation, const amt_axis* xaxis,
                  const amt_axis* yaxis, int lon_wrap, uint64_t* acc);
/* Mean + layout of auromat/resample.py:339-351 and :128-136: for out row r (lat descending) and
 * column c: cell (ix=c, iy=ny-1-r).  mean: (ny, nx, nchan+1) float64, NaN where count == 0;
 * out_img (optional): (ny, nx, nchan) of img_dtype, round-half-even of the mean (0 where empty);
 * out_mask (optional): (ny, nx) uint8, 1 where count == 0; out_count (optional): (ny, nx) float64. */
int amt_bin_frame_finalize(amt_ctx* ctx, const uint64_t* acc, int32_t nx, int32_t ny, int32_t nchan,
                           int32_t img_dtype, double* mean, void* out_img, uint8_t* out_mask,
                           double* out_count);
/* Same on a window of a larger accumulator: acc has acc_nx x acc_ny cells per plane (the superset grid of
 * a fused amt_georef_frame launch); the output covers its cells [off_x, off_x+nx) x [off_y, off_y+ny). */
int amt_bin_frame_finalize_window(amt_ctx* ctx, const uint64_t* acc, int32_t acc_nx, int32_t acc_ny,
                                  int32_t off_x, int32_t off_y, int32_t nx, int32_t ny, int32_t nchan,
                                  int32_t img_dtype, double* mean, void* out_img, uint8_t* out_mask,
                                  double* out_count);
/* Same for float accumulators of amt_hist2d_accumulate: mean[k] = sums[k]/count, NaN where empty,
 * transposed + flipped to (ny, nx, nweights). */
int amt_hist2d_finalize_mean(amt_ctx* ctx, const double* count, const double* const* sums, int32_t nweights,
                             int32_t nx, int32_t ny, double* mean);

/* ---- resample(method='nearest') and the outside-outline masking -------------------- */

/* auromat/resample.py:301-327: scipy.interpolate.griddata((lat_c, lon_c)[valid], values, (latSpaceCenter[:,None],
 * lonSpaceCenter[None,:]), method='nearest'), i.e. for every grid centre the valid pixel centre at the smallest
 * Euclidean distance in the (lat, lon) plane in degrees.  Valid pixels as in amt_bin_frame (finite coordinates,
 * center_mask == 0, elev >= min_elevation unless that is -inf); lon_wrap != 0 searches with wrap_at(lon + 180, 180).
 * xaxis / yaxis: the uniform histogram axes of the output grid (amt_grid.xaxis / .yaxis: one bin per grid centre);
 * target_lat (ny, north -> south) / target_lon (nx): the grid centres themselves (device arrays);
 * target_mask: (ny, nx) uint8, 1 = do not search (e.g. outside the outline), may be NULL.
 * out_index: (ny, nx) int64, the flat index of the nearest pixel, -1 where masked or when there is no valid pixel.
 * Of several pixels at exactly the same distance the one with the lowest index wins (the k-d tree of the reference
 * returns an unspecified one).  Uses the context's workspace (about 8 bytes per pixel + 12 per grid cell). */
int amt_nearest_frame(amt_ctx* ctx, const double* lat_c, const double* lon_c, const double* elev,
                      const uint8_t* center_mask, int32_t height, int32_t width, double min_elevation,
                      const amt_axis* xaxis, const amt_axis* yaxis, int lon_wrap, const double* target_lat,
                      const double* target_lon, const uint8_t* target_mask, int64_t* out_index);
/* Values of the pixels chosen by amt_nearest_frame, laid out like the outputs of amt_bin_frame_finalize:
 * mean (optional): (n_targets, nchan+1) float64 image channels + elevation, NaN where index < 0;
 * out_img (optional): (n_targets, nchan) of img_dtype (0 where index < 0); out_mask (optional): 1 where index < 0. */
int amt_nearest_gather(amt_ctx* ctx, const int64_t* index, int64_t n_targets, const void* img, int32_t img_dtype,
                       int32_t nchan, const double* elev, double* mean, void* out_img, uint8_t* out_mask);
/* method='linear' and 'cubic' (reference resample.py:323-326: scipy.interpolate.griddata on a Delaunay triangulation of the valid
 * pixel centres) run on the exact triangulation below (amt_delaunay_*).  The lattice approximations of rounds 3-4
 * (amt_linear_gather, amt_cubic_gradients, amt_cubic_gather: the Gauss-reduced local lattice of the pixel grid in place of
 * Qhull's triangulation) were retired with ABI v6. */
/* ---- method='cubic' on the exact triangulation (round 5) ----
 * scipy.interpolate.griddata(method='cubic'), the reference's call (resample.py:323-326), is CloughTocher2DInterpolator on
 * scipy.spatial.Delaunay(points): Qhull's Delaunay triangulation, gradients from a Gauss-Seidel relaxation over its edges in
 * the order of the points, the Clough-Tocher element per triangle.
 *   amt_delaunay_create   HOST: the Delaunay triangulation of n >= 3 points xy (n, 2) (host memory; unique unless four points
 *                         are cocircular to the last bit: then Qhull's diagonal is not reproduced).  AMT_EINVAL when all
 *                         points are collinear.  Points that coincide with an earlier one are left out (amt_delaunay_sizes).
 *   amt_delaunay_create_threads   the same with the number of threads (<= 0: AMT_DELAUNAY_THREADS, else up to 16) and the number of
 *                         points from which on the build is parallel (<= 0: AMT_DELAUNAY_PARALLEL_MIN, else 200 000): vertical
 *                         strips triangulated side by side and joined at their seams (common tangents, the gap filled, Lawson
 *                         flips) — the same triangulation, the triangles in another order.
 *   amt_delaunay_build_info   info2[2]: strips built side by side (1: the sequential build), edge flips spent on joining them
 *   amt_delaunay_stats    stats4[4]: orientation tests that left double precision, of those exact zeros (three points collinear),
 *                         in-circle tests that left double precision, of those still undecided at 113 bits (four points
 *                         cocircular).  [1] = [3] = 0: no tie was met — the triangulation is the unique Delaunay triangulation of
 *                         the points, hence Qhull's.
 *   amt_delaunay_triangles        simplices (nt, 3) counter-clockwise and neighbours (nt, 3): the triangle opposite vertex k
 *                                 or -1 (scipy.spatial.Delaunay.simplices / .neighbors, in another order of triangles)
 *   amt_delaunay_vertex_neighbours   CSR (indptr (n + 1) int64, indices int32): Delaunay.vertex_neighbor_vertices, every vertex's
 *                                 list in increasing order (made on first request: this call or amt_delaunay_sizes' n_neighbours)
 *   amt_delaunay_slots    (ABI v6) the build's own triangle slots without a copy: *vertices (n_slots, 3) int32, counter-clockwise,
 *                         -1 = the vertex at infinity of a ghost triangle (one per hull edge), and *dead (n_slots) uint8, 1 = a
 *                         slot that holds no triangle; host pointers into the handle, valid until amt_delaunay_destroy.  A caller
 *                         that wants the vertex lists ON THE DEVICE uploads these two arrays and builds the lists there (every
 *                         finite triangle gives the directed edges v1 -> v2, v2 -> v0, v0 -> v1; an edge without its reverse is a
 *                         hull edge and gets it; sorted by (source, target) they are the CSR above) instead of waiting for the
 *                         host to make and copy them: auromat_amd.resample.cubic_exact
 *   amt_delaunay_locate   for m targets (m, 2): the vertices (m, 3) of the triangle that holds each (-1: outside the hull), the
 *                         centroids (m, 3, 2) of the triangles across its three edges and has_neighbour (m, 3)
 *   amt_cubic_gradients_csr   DEVICE: interpnd._estimate_gradients_2d_global in scipy's order — every channel relaxed until the
 *                         largest relative change of a sweep is below `tolerance` (scipy: 1e-6) or max_iterations (400)
 *                         sweeps; iterations[nchan] (host) receives the sweeps per channel (nchan <= 63).  xy (n, 2), values (n, nchan),
 *                         gradients (n, nchan, 2) on the device; the points must be in row-major pixel order and
 *                         row_start (n_rows + 1, device) give the first point of every pixel row (a wave walks a row).
 *   amt_cubic_eval        DEVICE: the element at the m targets -> out (m, nchan); NaN outside the hull. */
typedef struct amt_delaunay amt_delaunay;
int amt_delaunay_create(const double* xy, int64_t n, amt_delaunay** out);
int amt_delaunay_create_threads(const double* xy, int64_t n, int32_t threads, int64_t parallel_min, amt_delaunay** out);
int amt_delaunay_build_info(const amt_delaunay* d, int64_t* info2);
int amt_delaunay_destroy(amt_delaunay* d);
int amt_delaunay_sizes(const amt_delaunay* d, int64_t* n_triangles, int64_t* n_neighbours, int64_t* n_duplicates);
int amt_delaunay_stats(const amt_delaunay* d, int64_t* stats4);
int amt_delaunay_triangles(const amt_delaunay* d, int32_t* simplices, int32_t* neighbours);
int amt_delaunay_slots(const amt_delaunay* d, const int32_t** vertices, const uint8_t** dead, int64_t* n_slots);
int amt_delaunay_vertex_neighbours(const amt_delaunay* d, int64_t* indptr, int32_t* indices);
int amt_delaunay_locate(const amt_delaunay* d, const double* targets, int64_t m, int32_t* vertices, double* centroids,
                        uint8_t* has_neighbour);
int amt_cubic_gradients_csr(amt_ctx* ctx, const double* xy, int64_t n, const int64_t* indptr, const int32_t* indices,
                            const int64_t* row_start, int32_t n_rows, const double* values, int32_t nchan, double tolerance,
                            int32_t max_iterations, double* gradients, int32_t* iterations);
int amt_cubic_eval(amt_ctx* ctx, int64_t m, const double* targets, const int32_t* vertices, const double* centroids,
                   const uint8_t* has_neighbour, const double* xy, const double* values, const double* gradients, int32_t nchan,
                   double* out);
/* auromat/utils.py:58-74 pointsInsidePolygon = matplotlib.path.Path(polygon).contains_points(points): crossing test
 * with Agg's half-open edge rule; polygon: (n_vertices, 2) device doubles (x, y), closed implicitly. */
int amt_points_in_polygon(amt_ctx* ctx, const double* px, const double* py, int64_t n, const double* polygon,
                          int32_t n_vertices, uint8_t* out_inside);

/* ---- grid layout and the single-pass frame driver ------------------------------------------ */

/* Output grid of resample(method='mean') for a bounding box: auromat/resample.py:281-299 fixedGrid (global
 * alignment to +-90/+-180 with 1/pxPerDeg spacing), :220-241 (centres; first and last dropped) and
 * :330-334 + util/histogram.py:186,215-224 (histogram ranges, edges, right-edge rounding).  Pure host
 * arithmetic, bit-identical to the NumPy expressions (np.linspace, round, argmax); no GPU needed. */
typedef struct amt_grid {
    int32_t nx, ny;                      /* output cells along longitude / latitude */
    int32_t n_lat_nodes, n_lon_nodes;    /* fixedGrid's nLat, nLon */
    double lat_lo, lat_hi, lon_lo, lon_hi;   /* global grid nodes enclosing the box */
    double lat_step, lon_step;           /* centre spacing; lat_step < 0 (rows run north -> south) */
    double lat_center_first, lat_center_last, lon_center_first, lon_center_last;
    amt_axis xaxis, yaxis;               /* uniform histogram axes (edges == NULL); y edges ascend */
} amt_grid;
/* Returns AMT_OK, or AMT_EINVAL when the box yields no output cell (the reference asserts nLat, nLon > 1). */
int amt_grid_layout(double lat_px_per_deg, double lon_px_per_deg, double lat_min, double lat_max,
                    double lon_min, double lon_max, amt_grid* out);

/* Single-pass driver: georeference + mask by elevation + bounding box + grid + binned mean of one frame with
 * the binning fused into the georeferencing kernel (see amt_georef_out.bin_*), for geodetic grids (frames that straddle
 * the 180 deg discontinuity are binned with shifted longitudes, frames with a pole in view in coordinates rotated by
 * 90 deg about x: reference resample.py:176-218) and for MLat/MLT grids (resampleMLatMLT) likewise.  Separate calls per
 * stage so that frames can be software pipelined by one host thread:
 *   amt_pipe_coarse   enqueue the coarse bounding-box pre-pass (own high-priority stream), any time earlier
 *   amt_pipe_launch   wait for it, lay out the superset grid, zero the accumulators, launch the fused kernel
 *                     on the context's stream, start the copy of the exact bounding box
 *   amt_pipe_wait     wait for the exact box (the only host synchronisation), lay out the exact grid
 *   amt_pipe_finalize crop + finalise into arrays the caller sized from the grid amt_pipe_wait returned
 * status in amt_pipe_result: 0 = ready to finalise; 1 = this frame needs the general path (exact box outside the
 * superset, an extreme of a pole frame's box within 1e-6 cells of a grid node, a geodetic pole frame whose caller also
 * wants MLat / MLT arrays) — the coordinate
 * arrays and bbox are valid, nothing else;
 * 2 = no pixel above the elevation threshold (mapping.py:858-859 -> ValueError). */
typedef struct amt_pipe amt_pipe;
typedef struct amt_pipe_result {
    int32_t status;
    int32_t fused;          /* 1 when the fused kernel was launched for this frame */
    int32_t lon_wrapped;    /* 1: the frame straddles the 180 deg discontinuity; `grid` is laid out for longitudes
                             * shifted by 180 deg (wrap_at_180(lon + 180)) and the caller shifts the output
                             * coordinates back (reference resample.py:203-218,274-277) */
    int32_t edge_pixels;    /* pixels on a bin edge (right-most-edge rule) that were resolved separately */
    double bbox[8];         /* exact reduction of amt_georef_frame; [7] = 1 when a pole (of the grid's coordinates) is in
                             * view: [0..5] are then in the rotated coordinates the grid is laid out in (amt_georef_out.
                             * bin_pole), and the caller rotates the output coordinates back (resample.py:262-273) */
    amt_grid grid;          /* exact output grid (valid for status 0) */
} amt_pipe_result;
int amt_pipe_create(amt_ctx* ctx, amt_pipe** out_pipe);
int amt_pipe_destroy(amt_pipe* pipe);
int amt_pipe_coarse(amt_pipe* pipe, const amt_frame_params* p, double min_elevation, int magnetic);
/* Instead of the pre-pass: the caller supplies the estimate (bbox[0..6] as amt_georef_coarse_bbox reports them),
 * e.g. the exact box amt_pipe_wait returned for a neighbouring frame of the same sequence.  Consecutive frames
 * move by a fraction of the superset margin, and a frame whose exact box does not fit its superset grid takes
 * the general path anyway (status 1), so a poor estimate costs time, never correctness. */
int amt_pipe_coarse_hint(amt_pipe* pipe, const double* bbox, int magnetic);
/* out: arrays to write (lat .. mlt_c as in amt_georef_frame; bbox / bin_* fields are managed by the driver) and
 * out->altitude, the mapping altitude [km] (read when a pole is in view).
 * img: (height, width, 3) uint8 (img_dtype 1) or uint16 (2).  min_elevation: -inf disables the mask.
 * pole_in_view: 0 / 1 = the caller's decision, < 0 = decide from the camera model (is a pole of the mapping
 * shell imaged by a valid pixel; replaces geodesic.py:183 / mapping.py:705-721 for camera mappings).
 * magnetic != 0 (same value as given to amt_pipe_coarse): grid, bounding box and pole refer to (MLat, SM
 * longitude), i.e. resampleMLatMLT (resample.py:63-71, mapping.py:1519-1547); p->m_sm must be set. */
int amt_pipe_launch(amt_pipe* pipe, const amt_frame_params* p, const amt_georef_out* out, const void* img,
                    int32_t img_dtype, double min_elevation, double lat_px_per_deg, double lon_px_per_deg,
                    int pole_in_view, int magnetic);
/* The single-pass plan for caller-supplied corner directions — north_star's "(H+1) x (W+1) corner arrays" form of the
 * pipeline, reference astrometry.py:49-64,86-106 with any camera model (BaseAstrometryMapping over pixelDirection's
 * array; here DirectionArrayMapping): amt_pipe_coarse_dirs samples the direction array for the estimate (or
 * amt_pipe_coarse_hint), amt_pipe_launch_dirs runs amt_georef_frame_dirs with the binning fused in; amt_pipe_wait /
 * amt_pipe_finalize as above.  p: width, height, cam, a, b, a0, b0, m_geo, m_sm and fast_center = 1 are read (the TAN
 * block is not).  There is no camera model to project a pole through: pole_in_view 0 / 1 is the caller's decision (1: the
 * frame is not fused, amt_pipe_wait returns status 1), < 0 = unknown — then a frame whose estimated or exact box comes
 * within 5 deg of a pole of its grid's coordinates is handed back with status 1 and the caller decides from the corner
 * arrays (amt_bbox_corners counts the quads that wind around a pole). */
int amt_pipe_coarse_dirs(amt_pipe* pipe, const amt_frame_params* p, const double* corner_dirs, double min_elevation,
                         int magnetic);
int amt_pipe_launch_dirs(amt_pipe* pipe, const amt_frame_params* p, const double* corner_dirs, const amt_georef_out* out,
                         const void* img, int32_t img_dtype, double min_elevation, double lat_px_per_deg,
                         double lon_px_per_deg, int pole_in_view, int magnetic);
/* amt_pipe_launch_dirs for n <= AMT_PIPE_MAX_BATCH frames, one per driver, each with its own direction array, in ONE launch of
 * the big kernel (ABI v6; see amt_pipe_launch_many below). */
int amt_pipe_launch_dirs_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p, const double* const* corner_dirs,
                              const amt_georef_out* const* out, const void* const* img, int32_t img_dtype, double min_elevation,
                              double lat_px_per_deg, double lon_px_per_deg, int pole_in_view, int magnetic);
/* The same for n <= AMT_PIPE_MAX_BATCH frames, one per driver (all on one context), with ONE launch of the big
 * kernel when the frames are equally sized and take the same kernel variant (their constants sit side by side in
 * the kernel-argument segment; otherwise one launch each): the 14-17 us between two big kernels on a stream are
 * then paid once per n frames.  amt_pipe_wait / amt_pipe_finalize stay per driver. */
#define AMT_PIPE_MAX_BATCH 3
int amt_pipe_launch_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p,
                         const amt_georef_out* const* out, const void* const* img, int32_t img_dtype,
                         double min_elevation, double lat_px_per_deg, double lon_px_per_deg, int pole_in_view,
                         int magnetic);
/* The same with a resolution per frame (lat_px_per_deg[n], lon_px_per_deg[n]): what `resample(arcsecPerPx=...)` needs,
 * where px/deg follows from each frame's own bounding box (amt_plate_carree_resolution). */
int amt_pipe_launch_many_res(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p,
                             const amt_georef_out* const* out, const void* const* img, int32_t img_dtype,
                             double min_elevation, const double* lat_px_per_deg, const double* lon_px_per_deg,
                             int pole_in_view, int magnetic);
/* Box-first plan — `resample(mapping, arcsecPerPx=R)`, the reference's own call form (auromat/cli/convert.py:176-185,
 * test/mapping_test.py:24-42): plateCarreeResolution (resample.py:36-61) needs the frame's EXACT bounding box before a grid
 * can be laid out.  amt_pipe_launch_box enqueues the frame kernel with no output array, no image and no binning — ray,
 * shell, coordinates, elevation mask, box reduction only — on the context's stream; the following amt_pipe_wait returns
 * status 1 (or 2: no valid pixel) with bbox[0..6] = the exact reduction in (lat, lon), or with magnetic != 0 in (MLat,
 * SM longitude), never rotated, and bbox[7] = pole in view (camera model).  The caller derives px/deg
 * (amt_plate_carree_resolution on the BoundingBox of that reduction), hands the box back as the estimate
 * (amt_pipe_coarse_hint with bbox[7] = 0; a pole frame runs amt_pipe_coarse instead: its grid lives in rotated coordinates)
 * and launches the ordinary single-pass plan, whose superset grid then always contains the exact one. */
int amt_pipe_launch_box(amt_pipe* pipe, const amt_frame_params* p, double min_elevation, int magnetic);
int amt_pipe_launch_box_many(amt_pipe* const* pipes, int32_t n, const amt_frame_params* const* p, double min_elevation,
                             int magnetic);
/* auromat/resample.py:36-61 plateCarreeResolution(BoundingBox(latSouth, lonWest, latNorth, lonEast), arcsecPerPx) ->
 * (latPxPerDeg, lonPxPerDeg); geodesic.angularDistance (geographiclib's a12 in the reference) restated from Karney's
 * integral formulation for two points on one parallel.  Host arithmetic, no GPU.  A box wider than 180 deg is measured the
 * shorter way round, as the reference does (min(lons, 360 - lons)); for a box that goes all the way around (a pole in view)
 * that gives *lon_px_per_deg = 0, which no grid can be laid out for: the outputs are filled as the reference's function
 * returns them — (3600 / arcsec_per_px, 0) — and the status is AMT_EDOMAIN (ABI v6; AMT_OK before), so that a host that checks
 * the status cannot lay out a grid without columns (the reference fails downstream on `assert nLon > 1`,
 * resample.py:226-227).  AMT_EINVAL when the resolution is not positive or the box has no width. */
int amt_plate_carree_resolution(double lat_south, double lon_west, double lat_north, double lon_east, double arcsec_per_px,
                                double* lat_px_per_deg, double* lon_px_per_deg);
int amt_pipe_wait(amt_pipe* pipe, amt_pipe_result* result);
/* mean (ny,nx,4) f64, out_img (ny,nx,3) of img_dtype, out_mask (ny,nx) u8, out_count (ny,nx) f64: device
 * buffers for result->grid of the preceding amt_pipe_wait (any may be NULL).  The kernel runs on the driver's
 * own stream so that it does not sit between two frames' big kernels on the context's stream. */
int amt_pipe_finalize(amt_pipe* pipe, double* mean, void* out_img, uint8_t* out_mask, double* out_count);
/* The same for the n <= AMT_PIPE_MAX_BATCH frames of one launch (arrays of n pointers each; all four kinds of output
 * are required here): ONE kernel finalises all of them and folds the on-edge pixels in, instead of two launches per
 * frame — the host side of the frame loop is what bounds short sequences and the grids-only mode. */
int amt_pipe_finalize_many(amt_pipe* const* pipes, int32_t n, double* const* mean, void* const* out_img,
                           uint8_t* const* out_mask, double* const* out_count);
/* The stream (hipStream_t) amt_pipe_finalize enqueues on.  A host whose allocator is stream ordered (PyTorch's caching
 * allocator, hipMallocAsync pools) must allocate the four output buffers FOR THIS STREAM: memory handed out for another
 * stream may still be in use by work queued there (the driver's finalise kernel does not wait for the context's
 * stream, that is its point), and the kernel would write into it too early. */
int amt_pipe_finalize_stream(amt_pipe* pipe, void** stream);
/* Orders the context's stream behind the last amt_pipe_finalize (no-op when it already completed): call it
 * before work on the context's stream — or, after amt_ctx_synchronize, the host — reads the outputs. */
int amt_pipe_join(amt_pipe* pipe);
/* The two-pass plan on the driver's streams, for frames amt_pipe_wait hands back with status 1 and for sequences that ask for
 * it (reference resample.py:159-279 as two steps: the coordinate arrays, then `_resampleCenterData`):
 *   amt_pipe_set_plan(pipe, 1)     the big kernel never bins (it writes the coordinate arrays of amt_georef_out and the box);
 *   amt_pipe_general_layout        after amt_pipe_wait: lays out the exact grid from the frame's box — result->grid,
 *                                  lon_wrapped, status 0 — unless the frame needs what only the caller has (a pole in view,
 *                                  exact centres, an MLat / MLT grid, no coordinate arrays): then status stays 1;
 *   amt_pipe_general_finalize      zeroes the accumulators, bins the frame's arrays (amt_bin_frame) and finalises into the
 *                                  caller's arrays (sized from result->grid), on the context's stream. */
int amt_pipe_set_plan(amt_pipe* pipe, int two_pass);
int amt_pipe_general_layout(amt_pipe* pipe, amt_pipe_result* result);
int amt_pipe_general_finalize(amt_pipe* pipe, double* mean, void* out_img, uint8_t* out_mask, double* out_count);

/* ---- native sequence runner ---------------------------------------------------------------------------------------
 * The per-frame loop of a sequence (reference mapping/spacecraft.py:326-332 `map(getMapping, ...)` followed by
 * cli/convert.py:178-185 `resample`) as ONE call: for every frame the host scalars (ephemeris seconds, the cxform
 * matrices J2000 -> GEO / SM with the IGRF dipole, the WCS Euler matrix: reference transform.py:491-696, wcs.py:133-139),
 * the estimate of its bounding box (the boxes of the frames finished before it, where the sequence is coherent; else the
 * coarse pre-pass), the launch (up to AMT_PIPE_MAX_BATCH frames per launch of the big kernel), the wait for its exact box,
 * the grid layout and the finalise kernel — software pipelined over `n_slots` frame slots by this one host thread.
 * Results: frame k's mean (ny, nx, 4) and count (ny, nx) lie one after the other at grids + grid_offset — consecutive
 * frames back to back, which is the payload of the gather's wire format (amt_seq_pack) without a copy —, its rounded
 * image (ny, nx, 3) and mask (ny, nx) at images + image_offset (256-byte aligned).
 * status per frame: 0 done (two_pass = 1: by the two-pass plan, which frames the single-pass plan hands back take when their
 * slot has coordinate arrays); 1 the frame needs the caller's general path (a pole in view that the pole plan does not cover,
 * ...; nothing of it is in the arenas); 2 no pixel above the elevation threshold;
 * 3 the arenas are full (this and the later frames were not processed); 4 (arcsec_per_px only) a pole of the grid's
 * coordinates is in view, for which plateCarreeResolution has no longitude resolution (reference resample.py:47-61: the box
 * goes all around; the reference fails on such a frame).
 * A frame the single-pass plan hands back although its launch was fused (its exact box does not fit the superset grid of a
 * poor estimate, the date line judged differently) is launched once more with its exact box as the estimate before it
 * takes any other path.
 * arcsec_per_px > 0 — `resample(mapping, arcsecPerPx=R)`, what the reference's auromat-convert runs (cli/convert.py:176-185) —:
 * every frame's px/deg follows from its own bounding box (amt_plate_carree_resolution; lat / lon_px_per_deg of the config are
 * ignored, the frame's pair is in its result): the box-first plan (amt_pipe_launch_box), the box pass of a batch two batches
 * ahead of its single-pass launch; needs n_slots >= 3 * batch. */
typedef struct amt_run amt_run;
typedef struct amt_run_config {
    int32_t width, height;
    int32_t img_dtype;            /* 1 = uint8, 2 = uint16; (height, width, 3) */
    int32_t fast_center;          /* BaseAstrometryMapping.fastCenterCalculation */
    int32_t magnetic;             /* grids in (MLat, SM longitude): resampleMLatMLT */
    int32_t batch;                /* frames per launch, 1 .. AMT_PIPE_MAX_BATCH */
    int32_t use_hints;            /* 0: coarse pre-pass for every frame */
    int32_t n_slots;              /* frame slots, >= 2 * batch */
    int32_t two_pass;             /* 1: the two-pass plan for every frame (needs the slots' lat_c / lon_c / elev arrays) */
    int32_t reserved_;
    double altitude;              /* mapping shell [km] of frames that name none */
    double min_elevation;         /* maskedByElevation; -inf disables */
    double lat_px_per_deg, lon_px_per_deg;
    const amt_georef_out* slots;  /* n_slots blocks: the per-pixel arrays each slot's frames write (NULL = not written);
                                   * bbox / bin_* fields are managed by the runner */
    double arcsec_per_px;         /* > 0: a resolution per frame from its own bounding box (see above); 0: the fixed pair */
} amt_run_config;
typedef struct amt_run_frame {
    double crval[2], crpix[2], cd[4], lonpole;   /* the TAN WCS cards (LATPOLE = 0) */
    double cam[3];                /* cameraPosGCRS [km] */
    double jd;                    /* photo time, UTC Julian date as ONE double (astropy Time(...).jd, transform.py:529) */
    double altitude;              /* mapping shell [km]; <= 0: the config's */
    const void* img;              /* device, (height, width, 3) of img_dtype; read until the call returns */
    /* ABI v6: with img == NULL, the image in PAGE-LOCKED host memory (hipHostMalloc / hipHostRegister / torch pin_memory; same
     * layout; read until amt_run_end returns).  The runner sends the rows that can be binned (amt_georef_image_rows) to a device
     * buffer of its own per slot on a copy stream of its own at once — the frame's launch comes one batch later and waits for
     * them —; amt_run_result.uploaded_bytes says how many bytes that were.  What the reference's loop does with the image it
     * has just read from disk (cli/convert.py:178-216, mapping/spacecraft.py:326-332). */
    const void* img_host;
} amt_run_frame;
typedef struct amt_run_result {
    int32_t status, slot;
    int32_t ny, nx;
    int32_t contains_pole;        /* grid laid out in coordinates rotated by +90 deg about x (resample.py:176-201) */
    int32_t lon_wrapped;          /* grid laid out in longitudes shifted by 180 deg (resample.py:203-218) */
    int32_t hinted;               /* 1: no coarse pre-pass ran for this frame */
    int32_t edge_pixels;
    int32_t two_pass;             /* 1: binned by the separate pass (amt_pipe_general_*), 0: by the fused kernel */
    int32_t reserved_;
    int64_t grid_offset;          /* doubles */
    int64_t image_offset;         /* bytes */
    double bbox[8];               /* amt_pipe_result.bbox */
    double altitude;
    amt_grid grid;
    amt_frame_params params;      /* what the frame was computed with */
    double lat_px_per_deg, lon_px_per_deg;   /* the resolution the frame was binned at */
    int32_t retried, reserved2_;  /* retried = 1: launched a second time with its exact box as the estimate */
    int64_t uploaded_bytes;       /* amt_run_frame.img_host: bytes of the image that crossed the link (0 for a device image) */
} amt_run_result;
/* The host scalars of one frame (no GPU call).  AMT_EINVAL: date outside the IGRF table with want_sm. */
int amt_frame_params_from_wcs(const amt_run_frame* frame, int32_t width, int32_t height, int32_t fast_center,
                              double altitude, int32_t want_sm, amt_frame_params* out);
int amt_run_create(amt_ctx* ctx, const amt_run_config* config, amt_run** out_run);
int amt_run_destroy(amt_run* run);
/* Processes frames[0 .. n) on the context's stream (+ the drivers' own streams); on return the context's stream is
 * ordered behind everything the call enqueued.  grids: device, grids_capacity doubles; images: device, images_capacity
 * bytes; both must stay untouched by other streams during the call.  results: n records (host).  frames_done (optional):
 * how many leading frames were processed (< n only with status 3).  The box hints carry over from call to call. */
int amt_run_process(amt_run* run, const amt_run_frame* frames, int32_t n, double* grids, int64_t grids_capacity,
                    void* images, int64_t images_capacity, amt_run_result* results, int32_t* frames_done);
/* The same call in pieces, for a host that produces its frames one by one (reading headers, decoding images): begin
 * with the arenas and room for max_frames results, push the frames in order — a push prepares its frame and, when a batch
 * is complete, finishes the older batch in flight and launches the new one, so the GPU starts after the first push —,
 * end to finish what is in flight and order the context's stream behind it.  `frame` is copied. */
int amt_run_begin(amt_run* run, double* grids, int64_t grids_capacity, void* images, int64_t images_capacity,
                  amt_run_result* results, int32_t max_frames);
int amt_run_push(amt_run* run, const amt_run_frame* frame);
int amt_run_end(amt_run* run, int32_t* frames_done);
/* Forget the box hints (the next frame gets a coarse pre-pass). */
int amt_run_reset_hints(amt_run* run);

/* ---- sequences over several GPUs: packing of per-frame grids for the gather ------------------------------------
 * Whole frames are independent (reference mapping/spacecraft.py:326-332 iterates them with a plain `map`,
 * cli/convert.py:178-185), so a sequence shards by frame, one process per GPU, and only the per-frame output grids
 * travel: each rank packs its grids into ONE device buffer, the ranks exchange (frames, payload length) and the
 * root gathers the buffers padded to the longest (ncclAllGather of 2 int64 + ncclGather / grouped send-recv over
 * xGMI; python: auromat_amd/sequence.py on torch.distributed).  These three entry points give a non-Python host
 * the same wire format:
 *   buffer = [ max_frames x AMT_SEQ_DESC_LEN doubles | payload ],  payload per frame = mean (ny, nx, nc) then
 *   count (ny, nx), float64;  descriptor = ny, nx, nc, lat of the first row's centres, lon of the first column's
 *   centres, dlat, dlon, frame index, contains_pole, contains_discontinuity, altitude [km], magnetic.
 * A frame that straddles the 180 deg discontinuity has its grid laid out in longitudes shifted by 180 deg, a pole
 * frame in coordinates rotated by +90 deg about x at `altitude` (reference resample.py:176-218,262-277); the two
 * flags tell the receiver.  ny = 0 marks a frame without any valid pixel (the reference raises ValueError there,
 * mapping.py:858-859). */
#define AMT_SEQ_DESC_LEN 12
typedef struct amt_seq_frame {
    int32_t ny, nx, nc;            /* grid rows, columns, planes of `mean` (image channels + elevation) */
    int32_t index;                 /* position of the frame in the whole sequence */
    double lat0, lon0, dlat, dlon;
    int32_t contains_pole, contains_discontinuity, magnetic, reserved;
    double altitude;
    const double* mean;            /* pack: device pointers; unpack: pointers INTO the host buffer */
    const double* count;
} amt_seq_frame;
/* Payload length in doubles of n frames (host arithmetic only). */
int amt_seq_payload_size(const amt_seq_frame* frames, int32_t n, int64_t* n_doubles);
/* Writes the descriptors of the n frames and copies their grids (device to device, on the context's stream) into
 * `buffer` (device, capacity_doubles >= max_frames * AMT_SEQ_DESC_LEN + payload size; max_frames >= n is the
 * largest frame count of any rank, so that all ranks' payloads start at the same offset).  The unused tail of
 * the descriptor table is zeroed. */
int amt_seq_pack(amt_ctx* ctx, const amt_seq_frame* frames, int32_t n, int32_t max_frames, double* buffer,
                 int64_t capacity_doubles);
/* Splits one rank's gathered buffer, already in HOST memory, into frames: n_frames descriptors are read,
 * frames with ny = 0 are skipped; returns the number of entries written to `out` (at most capacity) in *n_out. */
int amt_seq_unpack(const double* host_buffer, int64_t n_doubles, int32_t n_frames, int32_t max_frames,
                   amt_seq_frame* out, int32_t capacity, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* AUROMAT_HIP_H */
