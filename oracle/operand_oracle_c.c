/* operand_oracle_c.c — C restatement (OpenMP over cells) of the three consumer-side steps of oracle/operand_oracle.py for the
 * eps / Mandel operand: strain at the quadrature points, internal force, matrix-free tangent action.
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline legs): never linked into or imported by the
 * product package. It exists so that the CPU baseline of the device-resident Newton iteration is compiled, threaded code — the
 * reference's own path for these steps is compiled code too (DOLFINx / FFCx kernels behind fem.Expression.eval and
 * assemble_vector) — and not the speed of a NumPy checker.
 *
 * Reference: evaluate_operands -> fem.Expression(eps(Du), points).eval (src/dolfinx_external_operator/external_operator.py:386-402;
 * operand of doc/demo/demo_plasticity_von_mises.py:225-227), the residual form inner(sigma, eps(v)) dx (:266) and the Jacobian form
 * inner(C_tang : eps(du), eps(v)) dx (:378-391) assembled by DOLFINx (external_operator.py:463-486 builds the latter). The
 * arithmetic of those third-party calls is the textbook push-forward; it is restated here as oracle/operand_oracle.py states it
 * (eval_operand :22-51, operand_adjoint :61-97, tangent_apply :100-106) and pinned against that file in tests/test_oracle_golden.py.
 * Parity against DOLFINx itself: unpinned (see the header of operand_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <omp.h>

#define MAXG 3

typedef struct {
    int gdim, nd, ng, nq, x_cols; /* x_cols: row length of x (DOLFINx stores 3 columns also in 2-D) */
    const int32_t* dofmap;        /* [nc][nd] */
    const int32_t* geom_dofmap;   /* [nc][ng] */
    const double* x;              /* [n_x][x_cols] */
    const double* dphi;           /* [nq][nd][gdim] */
    const double* dpsi;           /* [nq][ng][gdim] */
    const double* w;              /* [nq] (adjoint / tangent only) */
} oracle_mesh;

/* K = J^-1 with J[j][k] = sum_v X_v[j] dpsi[q][v][k]; returns det J */
static double inverse_jacobian(const oracle_mesh* m, int64_t c, int q, double K[MAXG][MAXG]) {
    const int G = m->gdim;
    double J[MAXG][MAXG] = {{0}};
    for (int v = 0; v < m->ng; ++v) {
        const double* X = m->x + (int64_t)m->geom_dofmap[c * m->ng + v] * m->x_cols;
        const double* dp = m->dpsi + ((int64_t)q * m->ng + v) * G;
        for (int j = 0; j < G; ++j)
            for (int k = 0; k < G; ++k) J[j][k] += X[j] * dp[k];
    }
    if (G == 2) {
        const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        K[0][0] = J[1][1] / det; K[0][1] = -J[0][1] / det; K[1][0] = -J[1][0] / det; K[1][1] = J[0][0] / det;
        return det;
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    K[0][0] = c00 / det; K[1][0] = c01 / det; K[2][0] = c02 / det;
    K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det; K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det; K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det; K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    return det;
}

/* g[i][j] = du_i / dx_j at point q of cell c (operand_oracle.py:33-35) */
static void gradient(const oracle_mesh* m, int64_t c, int q, const double* u, const double K[MAXG][MAXG], double g[MAXG][MAXG]) {
    const int G = m->gdim;
    double gref[MAXG][MAXG] = {{0}};
    for (int a = 0; a < m->nd; ++a) {
        const double* ua = u + (int64_t)m->dofmap[c * m->nd + a] * G;
        const double* dp = m->dphi + ((int64_t)q * m->nd + a) * G;
        for (int i = 0; i < G; ++i)
            for (int k = 0; k < G; ++k) gref[i][k] += ua[i] * dp[k];
    }
    for (int i = 0; i < G; ++i)
        for (int j = 0; j < G; ++j) {
            double s = 0.0;
            for (int k = 0; k < G; ++k) s += gref[i][k] * K[k][j];
            g[i][j] = s;
        }
}

static void mandel(int G, const double g[MAXG][MAXG], double* e) { /* operand_oracle.py:43-48 */
    const double r = sqrt(2.0) * 0.5;
    if (G == 2) {
        e[0] = g[0][0]; e[1] = g[1][1]; e[2] = 0.0; e[3] = r * (g[0][1] + g[1][0]);
    } else {
        e[0] = g[0][0]; e[1] = g[1][1]; e[2] = g[2][2];
        e[3] = r * (g[0][1] + g[1][0]); e[4] = r * (g[0][2] + g[2][0]); e[5] = r * (g[1][2] + g[2][1]);
    }
}

#define MAXND 64 /* nodes per cell the element vector on the stack holds */

/* fe[a][i] += w |det J| sum_j gh[i][j] dphi_a/dx_j with gh the dual tensor of the Mandel vector s (operand_oracle.py:85-96): the
 * cell's element vector, added to the dof vector once per cell (add_element_vector), as an assembler does */
static void scatter(const oracle_mesh* m, int q, const double K[MAXG][MAXG], double det, const double* s, double* fe) {
    const int G = m->gdim;
    const double r = sqrt(2.0) * 0.5, scale = m->w[q] * fabs(det);
    double gh[MAXG][MAXG] = {{0}};
    if (G == 2) {
        gh[0][0] = s[0]; gh[1][1] = s[1]; gh[0][1] = gh[1][0] = r * s[3];
    } else {
        gh[0][0] = s[0]; gh[1][1] = s[1]; gh[2][2] = s[2];
        gh[0][1] = gh[1][0] = r * s[3]; gh[0][2] = gh[2][0] = r * s[4]; gh[1][2] = gh[2][1] = r * s[5];
    }
    for (int a = 0; a < m->nd; ++a) {
        const double* dp = m->dphi + ((int64_t)q * m->nd + a) * G;
        double gp[MAXG];
        for (int j = 0; j < G; ++j) {
            double t = 0.0;
            for (int k = 0; k < G; ++k) t += dp[k] * K[k][j];
            gp[j] = t;
        }
        for (int i = 0; i < G; ++i) {
            double t = 0.0;
            for (int j = 0; j < G; ++j) t += gh[i][j] * gp[j];
            fe[a * G + i] += scale * t;
        }
    }
}

static void add_element_vector(const oracle_mesh* m, int64_t c, const double* fe, double* out) {
    const int G = m->gdim;
    for (int a = 0; a < m->nd; ++a) {
        double* o = out + (int64_t)m->dofmap[c * m->nd + a] * G;
        for (int i = 0; i < G; ++i) {
#pragma omp atomic
            o[i] += fe[a * G + i];
        }
    }
}

/* e [nc][nq][d] = eps(u) in Mandel form, d = 4 (gdim 2) or 6 */
int oracle_operand_eps(const oracle_mesh* m, int64_t nc, const double* u, double* e, int nthreads) {
    if (!m || !u || !e || (m->gdim != 2 && m->gdim != 3)) return -1;
    const int D = m->gdim == 2 ? 4 : 6;
#pragma omp parallel for num_threads(nthreads < 1 ? 1 : nthreads) schedule(static)
    for (int64_t c = 0; c < nc; ++c)
        for (int q = 0; q < m->nq; ++q) {
            double K[MAXG][MAXG], g[MAXG][MAXG];
            (void)inverse_jacobian(m, c, q, K);
            gradient(m, c, q, u, K, g);
            mandel(m->gdim, g, e + (c * m->nq + q) * D);
        }
    return 0;
}

/* out [n_nodes][gdim] += sum_q w |det J| B^T S (S [nc][nq][d]) */
int oracle_operand_eps_adjoint(const oracle_mesh* m, int64_t nc, const double* S, double* out, int nthreads) {
    if (!m || !S || !out || !m->w || (m->gdim != 2 && m->gdim != 3) || m->nd > MAXND) return -1;
    const int D = m->gdim == 2 ? 4 : 6;
#pragma omp parallel for num_threads(nthreads < 1 ? 1 : nthreads) schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        double fe[MAXND * MAXG] = {0};
        for (int q = 0; q < m->nq; ++q) {
            double K[MAXG][MAXG];
            const double det = inverse_jacobian(m, c, q, K);
            scatter(m, q, K, det, S + (c * m->nq + q) * D, fe);
        }
        add_element_vector(m, c, fe, out);
    }
    return 0;
}

/* out += K v, K = sum_q w |det J| B^T C_tang B never formed (C_tang [nc][nq][d][d]) */
int oracle_tangent_apply(const oracle_mesh* m, int64_t nc, const double* C_tang, const double* v, double* out, int nthreads) {
    if (!m || !C_tang || !v || !out || !m->w || (m->gdim != 2 && m->gdim != 3) || m->nd > MAXND) return -1;
    const int D = m->gdim == 2 ? 4 : 6;
#pragma omp parallel for num_threads(nthreads < 1 ? 1 : nthreads) schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        double fe[MAXND * MAXG] = {0};
        for (int q = 0; q < m->nq; ++q) {
            double K[MAXG][MAXG], g[MAXG][MAXG], e[6], t[6];
            const double det = inverse_jacobian(m, c, q, K);
            gradient(m, c, q, v, K, g);
            mandel(m->gdim, g, e);
            const double* Cq = C_tang + (c * m->nq + q) * (int64_t)(D * D);
            for (int r = 0; r < D; ++r) {
                double acc = 0.0;
                for (int s = 0; s < D; ++s) acc += Cq[r * D + s] * e[s];
                t[r] = acc;
            }
            scatter(m, q, K, det, t, fe);
        }
        add_element_vector(m, c, fe, out);
    }
    return 0;
}
