"""A SECOND, independently written restatement of the reference's arithmetic on this path -- pure Python, one scalar at a time,
written from the Rust sources with their own structure kept (small vector / matrix values with nalgebra's evaluation order instead
of the scalars oracle/ref_literal.c expands everything into).  TEST INFRASTRUCTURE ONLY: tests/test_ref_python_crosscheck.py lays it
beside ref_literal.c on random inputs and asks for identical bits, so that a slip of transcription in either restatement shows up as a
disagreement instead of hiding behind a tolerance.  It pins nothing to the reference's own binary (there is none here: PARITY STAYS
UNPINNED); it makes "the restatement follows the source" a checked statement rather than a read one.

Followed (relative to /root/reference/crates/):
  rust_robotics_localization/src/particle_filter.rs          :279-296 predict, :316-329 update, :476-479 gauss_likelihood,
      :426-439 normalize_weights, :416-423 calc_n_eff, :441-473 resample_particles, :382-396 compute_estimate, :398-413 compute_covariance
  rust_robotics_localization/src/monte_carlo_localization.rs :328-336 cumulative weights with the last forced to 1, :387-392 sample_index
  rust_robotics_slam/src/fastslam1.rs                         :70-77 motion_model, :80-89 normalize_angle, :92-99 observation_model,
      :102-110 compute_jacobian, :123-137 predict_particle, :140-183 update_landmark, :186-193 compute_neff, :196-203 normalize_weights,
      :205-234 resample, :237-266 fastslam_update, :269-274 get_best_particle
  rust_robotics_slam/src/fastslam2.rs                         :105-118 motion_jacobian, :122-148 observation model and its two Jacobians,
      :173-216 compute_proposal, :219-239 sample_pose_with_rng, :242-280 update_landmark_and_weight, :331-374 fastslam2_update_with_rng
  rust_robotics_localization/src/monte_carlo_localization.rs :322-385 resample_adaptive, kld_required_particles, quantize_particle
nalgebra 0.33.2 (not under /root/reference; restated from its published source): a product of small static matrices is evaluated
column by column of the right-hand side, every entry summed over k in ascending order, no fused multiply-add (base/blas.rs gemm ->
gemv -> axcpy); Matrix2::try_inverse is `det = m11 m22 - m21 m12; None if det == 0; [m22, -m12; -m21, m11] / det` (linalg/inverse.rs);
Matrix2::determinant is the same `m11 m22 - m21 m12` (linalg/determinant.rs).
Python floats are IEEE doubles and `math` calls the C library's sin / cos / exp / sqrt / atan2 -- the functions ref_literal.c links
(sine-and-cosine pairs go through its sincos, as compiled code's do: see `sincos` below)."""
import ctypes
import ctypes.util
import math

PI = math.pi  # std::f64::consts::PI

# Where the source takes the sine AND the cosine of one argument (particle_filter.rs:292-293, fastslam1.rs:73-74, :146-147) an
# optimising compiler makes ONE call of the C library's sincos of the pair -- gcc does for ref_literal.c, LLVM does for the Rust -- and
# glibc's sincos is not bit for bit its sin and its cos (seen here: one last-bit difference in ~10^3 arguments).  The pair is therefore
# taken through sincos here as well; a lone sin or cos stays math.sin / math.cos.
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sincos.restype = None
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]


def sincos(x):
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(x, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


# ------------------------------------------------------------------ tiny nalgebra: row-major lists, nalgebra's order of evaluation
def mat_mul(a, b):
    """a (r x m) * b (m x c): entry (i, j) = sum over k ascending of a[i][k] * b[k][j], starting from the k = 0 product"""
    r, m, c = len(a), len(b), len(b[0])
    out = [[0.0] * c for _ in range(r)]
    for j in range(c):  # column by column of the right-hand side (gemm -> gemv)
        for i in range(r):
            acc = a[i][0] * b[0][j]  # axcpy with beta == 0: the first product is stored, not added to zero
            for k in range(1, m):
                acc = a[i][k] * b[k][j] + acc
            out[i][j] = acc
    return out


def mat_add(a, b):
    return [[x + y for x, y in zip(ra, rb)] for ra, rb in zip(a, b)]


def mat_sub(a, b):
    return [[x - y for x, y in zip(ra, rb)] for ra, rb in zip(a, b)]


def transpose(a):
    return [list(col) for col in zip(*a)]


def identity2():
    return [[1.0, 0.0], [0.0, 1.0]]


def determinant2(m):
    return m[0][0] * m[1][1] - m[1][0] * m[0][1]


def try_inverse2(m):
    det = m[0][0] * m[1][1] - m[1][0] * m[0][1]
    if det == 0.0:
        return None
    return [[m[1][1] / det, -m[0][1] / det], [-m[1][0] / det, m[0][0] / det]]


# ------------------------------------------------------------------ particle_filter.rs
def gauss_likelihood(x, sigma):
    coeff = 1.0 / math.sqrt(2.0 * PI * (sigma * sigma))  # sigma.powi(2)
    return coeff * math.exp(-(x * x) / (2.0 * (sigma * sigma)))


def pf_predict(particles, control, dt, v_noise, yaw_noise):
    """particles: list of [x, y, yaw, v, w]; the two noise lists stand for the Normal samples (0.0 where the sigma is 0)"""
    for p, nv, nw in zip(particles, v_noise, yaw_noise):
        v_noisy = control[0] + nv
        yaw_rate_noisy = control[1] + nw
        sin_yaw, cos_yaw = sincos(p[2])
        p[0] += v_noisy * cos_yaw * dt
        p[1] += v_noisy * sin_yaw * dt
        p[2] += yaw_rate_noisy * dt
        p[3] = v_noisy


def pf_update_raw(particles, observations, range_noise):
    for p in particles:
        w = 1.0
        for d_obs, lx, ly in observations:
            dx = p[0] - lx
            dy = p[1] - ly
            d_pred = math.sqrt(dx * dx + dy * dy)
            diff = d_obs - d_pred
            w *= gauss_likelihood(diff, range_noise)
        p[4] = w


def pf_normalize(particles):
    sum_w = 0.0
    for p in particles:
        sum_w += p[4]
    if sum_w > 0.0:
        for p in particles:
            p[4] /= sum_w
    else:
        uniform = 1.0 / float(len(particles))
        for p in particles:
            p[4] = uniform
    return sum_w


def pf_n_eff(particles):
    s2 = 0.0
    for p in particles:
        s2 += p[4] * p[4]
    return 1.0 / s2 if s2 > 0.0 else 0.0


def pf_estimate(particles):
    e = [0.0, 0.0, 0.0, 0.0]
    for p in particles:
        for k in range(4):
            e[k] += p[4] * p[k]
    return e


def pf_covariance(particles, est):
    cov = [[0.0] * 4 for _ in range(4)]
    for p in particles:
        dx = [p[k] - est[k] for k in range(4)]
        wdx = [p[4] * d for d in dx]  # particle.w * dx
        outer = [[wdx[r] * dx[c] for c in range(4)] for r in range(4)]  # (w dx) dx^T
        cov = mat_add(cov, outer)
    return cov


def pf_resample_indices(weights, draws):
    """particle_filter.rs:441-473: first i with r <= c[i], index 0 when there is none"""
    cumulative = []
    cum_sum = 0.0
    for w in weights:
        cum_sum += w
        cumulative.append(cum_sum)
    out = []
    for r in draws:
        index = 0
        for i, cum_w in enumerate(cumulative):
            if r <= cum_w:
                index = i
                break
        out.append(index)
    return out


def mcl_resample_indices(weights, draws):
    """monte_carlo_localization.rs:328-336, 387-392: the last cumulative weight is 1.0 by decree, the LAST index when nothing matches"""
    cumulative = []
    cum_sum = 0.0
    for w in weights:
        cum_sum += w
        cumulative.append(cum_sum)
    if cumulative:
        cumulative[-1] = 1.0
    out = []
    for r in draws:
        pos = None
        for i, w in enumerate(cumulative):
            if r <= w:
                pos = i
                break
        out.append(pos if pos is not None else len(cumulative) - 1)
    return out


# ------------------------------------------------------------------ fastslam1.rs
DT = 0.1
Q_SIM = [[0.3, 0.0], [0.0, 0.0305]]  # fastslam1.rs:22 (the literals, not (10 deg)^2 evaluated)
R_SIM = [[0.5, 0.0], [0.0, 0.0305]]  # fastslam1.rs:23


def normalize_angle(angle):
    a = angle
    while a > PI:
        a -= 2.0 * PI
    while a < -PI:
        a += 2.0 * PI
    return a


def motion_model(x, u, dt=DT):
    yaw = x[2]
    sin_yaw, cos_yaw = sincos(yaw)
    return [x[0] + u[0] * dt * cos_yaw, x[1] + u[0] * dt * sin_yaw, normalize_angle(x[2] + u[1] * dt)]


class Landmark:
    def __init__(self, x=0.0, y=0.0, cov=None):
        self.x, self.y = x, y
        self.cov = cov if cov is not None else [[1000.0, 0.0], [0.0, 1000.0]]


class Particle:
    def __init__(self, n_landmarks, weight=1.0 / 100.0):
        self.weight, self.x, self.y, self.yaw = weight, 0.0, 0.0, 0.0
        self.landmarks = [Landmark() for _ in range(n_landmarks)]

    def clone(self):
        q = Particle(0, self.weight)
        q.x, q.y, q.yaw = self.x, self.y, self.yaw
        q.landmarks = [Landmark(l.x, l.y, [row[:] for row in l.cov]) for l in self.landmarks]
        return q


def observation_model(particle, lm_id):
    lm = particle.landmarks[lm_id]
    dx = lm.x - particle.x
    dy = lm.y - particle.y
    d = math.sqrt(dx * dx + dy * dy)
    angle = normalize_angle(math.atan2(dy, dx) - particle.yaw)
    return [d, angle]


def compute_jacobian(particle, lm_id):
    lm = particle.landmarks[lm_id]
    dx = lm.x - particle.x
    dy = lm.y - particle.y
    d2 = dx * dx + dy * dy
    d = math.sqrt(d2)
    return [[dx / d, dy / d], [-dy / d2, dx / d2]]


def predict_particle(particle, u, n0, n1, q=Q_SIM, dt=DT):
    """n0, n1 stand for the two standard-normal samples of fastslam1.rs:129-130"""
    u_noisy = [u[0] + n0 * math.sqrt(q[0][0]), u[1] + n1 * math.sqrt(q[1][1])]
    pose = motion_model([particle.x, particle.y, particle.yaw], u_noisy, dt)
    particle.x, particle.y, particle.yaw = pose


def update_landmark(particle, z, lm_id, r, init_threshold=100.0):
    lm = particle.landmarks[lm_id]
    if lm.cov[0][0] > init_threshold:
        sin_b, cos_b = sincos(particle.yaw + z[1])
        lm.x = particle.x + z[0] * cos_b
        lm.y = particle.y + z[0] * sin_b
        return
    z_pred = observation_model(particle, lm_id)
    y = [[z[0] - z_pred[0]], [normalize_angle(z[1] - z_pred[1])]]  # a column
    h = compute_jacobian(particle, lm_id)
    s = mat_add(mat_mul(mat_mul(h, lm.cov), transpose(h)), r)
    s_inv = try_inverse2(s)
    if s_inv is None:
        s_inv = identity2()
    k = mat_mul(mat_mul(lm.cov, transpose(h)), s_inv)
    delta = mat_mul(k, y)
    lm.x += delta[0][0]
    lm.y += delta[1][0]
    lm.cov = mat_mul(mat_sub(identity2(), mat_mul(k, h)), lm.cov)
    det_s = determinant2(s)
    if det_s > 0.0:
        mahal = mat_mul(mat_mul(transpose(y), s_inv), y)
        likelihood = math.exp(-0.5 * mahal[0][0]) / (2.0 * PI * math.sqrt(det_s))
        particle.weight *= likelihood


def compute_neff(particles):
    s2 = 0.0
    for p in particles:
        s2 += p.weight * p.weight
    return 1.0 / s2 if s2 > 0.0 else 0.0


def normalize_weights(particles):
    s = 0.0
    for p in particles:
        s += p.weight
    if s > 0.0:
        for p in particles:
            p.weight /= s


def resample(particles, r0):
    """fastslam1.rs:205-234; r0 stands for the Uniform[0, 1/n) sample.  Returns (new particles, source indices)."""
    normalize_weights(particles)
    n = len(particles)
    cum_sum = [0.0] * (n + 1)
    for i, p in enumerate(particles):
        cum_sum[i + 1] = cum_sum[i] + p.weight
    r = r0
    j = 0
    new, idx = [], []
    for _ in range(n):
        while r > cum_sum[j + 1] and j < n - 1:
            j += 1
        q = particles[j].clone()
        q.weight = 1.0 / float(n)
        new.append(q)
        idx.append(j)
        r += 1.0 / float(n)
    return new, idx


def fastslam_update(particles, u, z, noise, r0, nth=100.0 / 1.5, r=R_SIM, q=Q_SIM, dt=DT, init_threshold=100.0):
    """fastslam1.rs:237-266; noise = per-particle (n0, n1); z = (d, angle, id) rows.  Returns (particles, fired, indices or None)."""
    for p, (n0, n1) in zip(particles, noise):
        predict_particle(p, u, n0, n1, q, dt)
    for d, angle, lm_id in z:
        for p in particles:
            update_landmark(p, [d, angle], int(lm_id), r, init_threshold)
    normalize_weights(particles)
    if compute_neff(particles) < nth:
        particles, idx = resample(particles, r0)
        return particles, True, idx
    return particles, False, None


def get_best_particle_index(particles):
    """Iterator::max_by keeps the LAST of equal maxima"""
    best = 0
    for i in range(1, len(particles)):
        a, b = particles[best].weight, particles[i].weight
        if not (b < a):  # partial_cmp(best, candidate) != Greater: the candidate replaces it
            best = i
    return best


# ------------------------------------------------------------------ monte_carlo_localization.rs: the KLD-adaptive resample
X_BIN_SIZE = 0.5  # monte_carlo_localization.rs:23-25
Y_BIN_SIZE = 0.5
YAW_BIN_SIZE = 15.0 * PI / 180.0


def _as_i32(v):
    """Rust's `f64 as i32`: truncation towards zero, saturating at the type's ends, NaN -> 0"""
    if v != v:
        return 0
    if v <= -2147483648.0:
        return -2147483648
    if v >= 2147483647.0:
        return 2147483647
    return int(v)


def _floor(v):
    """f64::floor: a float back (Python's math.floor makes an int and refuses NaN and the infinities)"""
    return v if v != v or abs(v) == math.inf else float(math.floor(v))


def quantize_particle(x, y, yaw):
    return (_as_i32(_floor(x / X_BIN_SIZE)), _as_i32(_floor(y / Y_BIN_SIZE)), _as_i32(_floor(yaw / YAW_BIN_SIZE)))


def kld_required_particles(k_bins, min_particles, max_particles, kld_epsilon, kld_z):
    """:367-378; `n.ceil() as usize` saturates (negative and NaN -> 0) before the clamp"""
    if k_bins <= 1:
        return min_particles
    k_minus_one = float(k_bins - 1)
    term = 1.0 - 2.0 / (9.0 * k_minus_one) + kld_z * math.sqrt(2.0 / (9.0 * k_minus_one))
    n = (k_minus_one / (2.0 * kld_epsilon)) * (term * term * term)  # term.powi(3)
    c = math.ceil(n) if n == n and abs(n) != math.inf else n
    as_usize = 0 if not (c > 0.0) else (2**64 - 1 if c >= 18446744073709551615.0 else int(c))
    return min(max(as_usize, min_particles), max_particles)


def resample_adaptive_indices(xs, ys, yaws, weights, draws, min_particles, max_particles, kld_epsilon, kld_z):
    """:322-365: the source index of every particle the loop keeps; draws[k] stands for the k-th `rng.random::<f64>()`"""
    if not weights:
        return []
    cumulative = []
    cum_sum = 0.0
    for w in weights:
        cum_sum += w
        cumulative.append(cum_sum)
    cumulative[-1] = 1.0
    bins = set()
    new = []
    required = min_particles
    while len(new) < max_particles:
        r = draws[len(new)]
        idx = len(cumulative) - 1
        for i, w in enumerate(cumulative):  # sample_index :387-392
            if r <= w:
                idx = i
                break
        bins.add(quantize_particle(xs[idx], ys[idx], yaws[idx]))
        required = max(required, kld_required_particles(len(bins), min_particles, max_particles, kld_epsilon, kld_z))
        new.append(idx)
        if len(new) >= min_particles and len(new) >= required:
            break
    return new


# ------------------------------------------------------------------ fastslam2.rs
# nalgebra 0.33.2 pieces this file needs beyond the 2 x 2 ones above, restated from the crate's published source (none of it is under
# /root/reference): Matrix3::try_inverse (linalg/inverse.rs, the cofactor form with its three shared minors), Cholesky::new + l()
# (linalg/cholesky.rs: column by column, col_j -= L[j][k] * col_k as an axpy over rows j.., None on a zero or negative pivot).
MOTION_COV = [[0.1, 0.0, 0.0], [0.0, 0.1, 0.0], [0.0, 0.0, 0.01]]  # fastslam2.rs:30


def try_inverse3(m):
    (m11, m12, m13), (m21, m22, m23), (m31, m32, m33) = m
    minor_m12_m23 = m22 * m33 - m32 * m23
    minor_m11_m23 = m21 * m33 - m31 * m23
    minor_m11_m22 = m21 * m32 - m31 * m22
    determinant = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22
    if determinant == 0.0:
        return None
    return [[minor_m12_m23 / determinant, (m13 * m32 - m33 * m12) / determinant, (m12 * m23 - m22 * m13) / determinant],
            [-minor_m11_m23 / determinant, (m11 * m33 - m31 * m13) / determinant, (m13 * m21 - m23 * m11) / determinant],
            [minor_m11_m22 / determinant, (m12 * m31 - m32 * m11) / determinant, (m11 * m22 - m21 * m12) / determinant]]


def cholesky_l(a):
    """The lower factor, or None where nalgebra's Cholesky::new gives up"""
    n = len(a)
    m = [row[:] for row in a]
    for j in range(n):
        for k in range(j):
            factor = -m[j][k]
            for i in range(j, n):  # col_j[j..] = factor * col_k[j..] + col_j[j..]
                m[i][j] = factor * m[i][k] + m[i][j]
        diag = m[j][j]
        if diag == 0.0 or not (diag >= 0.0):
            return None
        denom = math.sqrt(diag)
        m[j][j] = denom
        for i in range(j + 1, n):
            m[i][j] /= denom
    return [[m[i][j] if j <= i else 0.0 for j in range(n)] for i in range(n)]


def motion_jacobian(x, u, dt=DT):
    yaw, v = x[2], u[0]
    sin_yaw, cos_yaw = sincos(yaw)
    return [[1.0, 0.0, -v * dt * sin_yaw], [0.0, 1.0, v * dt * cos_yaw], [0.0, 0.0, 1.0]]


def fs2_observation_model(px, py, pyaw, lm_x, lm_y):
    dx = lm_x - px
    dy = lm_y - py
    d = math.sqrt(dx * dx + dy * dy)
    return [d, normalize_angle(math.atan2(dy, dx) - pyaw)]


def obs_jacobian_landmark(px, py, lm_x, lm_y):
    dx = lm_x - px
    dy = lm_y - py
    d2 = dx * dx + dy * dy
    d = math.sqrt(d2)
    return [[dx / d, dy / d], [-dy / d2, dx / d2]]


def obs_jacobian_pose(px, py, lm_x, lm_y):
    dx = lm_x - px
    dy = lm_y - py
    d2 = dx * dx + dy * dy
    d = math.sqrt(d2)
    return [[-dx / d, -dy / d, 0.0], [dy / d2, -dx / d2, -1.0]]


def fs2_is_initialized(lm):
    return lm.cov[0][0] < 100.0


def compute_proposal(particle, u, z, lm_id, r):
    """fastslam2.rs:173-216 -> (mean as a list of 3, covariance 3 x 3)"""
    lm = particle.landmarks[lm_id]
    pose = [particle.x, particle.y, particle.yaw]
    x_pred = motion_model(pose, u)
    g = motion_jacobian(pose, u)
    p_pred = mat_mul(mat_mul(g, MOTION_COV), transpose(g))
    if not fs2_is_initialized(lm):
        return x_pred, p_pred
    h_pose = obs_jacobian_pose(x_pred[0], x_pred[1], lm.x, lm.y)
    h_lm = obs_jacobian_landmark(x_pred[0], x_pred[1], lm.x, lm.y)
    q_obs = mat_add(mat_mul(mat_mul(h_lm, lm.cov), transpose(h_lm)), r)
    h_pose_t = transpose(h_pose)
    q_obs_inv = try_inverse2(q_obs)
    if q_obs_inv is None:
        q_obs_inv = identity2()
    p_pred_inv = try_inverse3(p_pred)
    if p_pred_inv is None:
        p_pred_inv = [[1e-6 if i == j else 0.0 * 1e-6 for j in range(3)] for i in range(3)]  # Matrix3::identity() * 1e-6
    p_post_inv = mat_add(p_pred_inv, mat_mul(mat_mul(h_pose_t, q_obs_inv), h_pose))
    p_post = try_inverse3(p_post_inv)
    if p_post is None:
        p_post = p_pred
    z_pred = fs2_observation_model(x_pred[0], x_pred[1], x_pred[2], lm.x, lm.y)
    innovation = [[z[0] - z_pred[0]], [normalize_angle(z[1] - z_pred[1])]]
    corr = mat_mul(mat_mul(mat_mul(p_post, h_pose_t), q_obs_inv), innovation)
    return [x_pred[k] + corr[k][0] for k in range(3)], p_post


def sample_pose(mean, cov, noise):
    """:219-239; noise stands for the three standard-normal samples, in the order they are drawn"""
    l = cholesky_l(cov)
    if l is None:
        l = [[math.sqrt(max(cov[i][i], 0.0)) if i == j else 0.0 for j in range(3)] for i in range(3)]
    ln = mat_mul(l, [[noise[0]], [noise[1]], [noise[2]]])
    return [mean[k] + ln[k][0] for k in range(3)]


def update_landmark_and_weight(particle, z, lm_id, r):
    """:242-280 -> the weight factor"""
    lm = particle.landmarks[lm_id]
    if not fs2_is_initialized(lm):
        sin_b, cos_b = sincos(particle.yaw + z[1])
        lm.x = particle.x + z[0] * cos_b
        lm.y = particle.y + z[0] * sin_b
        lm.cov = [[10.0, 0.0 * 10.0], [0.0 * 10.0, 10.0]]  # Matrix2::identity() * 10.0
        return 1.0
    z_pred = fs2_observation_model(particle.x, particle.y, particle.yaw, lm.x, lm.y)
    innovation = [[z[0] - z_pred[0]], [normalize_angle(z[1] - z_pred[1])]]
    h = obs_jacobian_landmark(particle.x, particle.y, lm.x, lm.y)
    s = mat_add(mat_mul(mat_mul(h, lm.cov), transpose(h)), r)
    s_inv = try_inverse2(s)
    if s_inv is None:
        s_inv = identity2()
    k = mat_mul(mat_mul(lm.cov, transpose(h)), s_inv)
    delta = mat_mul(k, innovation)
    lm.x += delta[0][0]
    lm.y += delta[1][0]
    lm.cov = mat_mul(mat_sub(identity2(), mat_mul(k, h)), lm.cov)
    det_s = determinant2(s)
    if det_s > 0.0:
        mahal = mat_mul(mat_mul(transpose(innovation), s_inv), innovation)
        return math.exp(-0.5 * mahal[0][0]) / (2.0 * PI * math.sqrt(det_s))
    return 1e-10


def fastslam2_update(particles, u, z, noise, draw_r0, nth=100.0 / 1.5, r=R_SIM, q=Q_SIM):
    """:331-374.  noise[p] = the normals particle p draws (three with an observation, two without); draw_r0() is called only when the
    gate fires, as the reference draws its uniform inside resample_with_rng.  Returns (particles, fired)."""
    for particle, nz in zip(particles, noise):
        if len(z):
            d, angle, lm_id = z[0]
            mean, cov = compute_proposal(particle, u, [d, angle], int(lm_id), r)
            pose = sample_pose(mean, cov, nz)
        else:
            u_noisy = [u[0] + nz[0] * math.sqrt(q[0][0]), u[1] + nz[1] * math.sqrt(q[1][1])]
            pose = motion_model([particle.x, particle.y, particle.yaw], u_noisy)
        particle.x, particle.y, particle.yaw = pose[0], pose[1], normalize_angle(pose[2])  # set_pose :78-82
        for d, angle, lm_id in z:
            particle.weight *= update_landmark_and_weight(particle, [d, angle], int(lm_id), r)
    normalize_weights(particles)
    if compute_neff(particles) < nth:
        particles, _ = resample(particles, draw_r0())
        return particles, True
    return particles, False
