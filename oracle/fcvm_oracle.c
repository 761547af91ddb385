/*
 * oracle/fcvm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the per-element hot path of the reference
 * (HarryvL/fcVM-workbench, "source code/fcVM.py").  It exists only to CHECK the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under
 * fcvm-workbench_amd/ may include, link, import or call it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py replays the reference's
 * Embankment_with_Ditch_Example (GNLN) and Bar buckling (GNLY, incl. buckling factors) analyses with
 * these routines and matches the reference's own result files (.out tables, .vtk nodal fields in
 * f64, stored in tests/golden/).
 *
 * Every routine cites the reference lines it restates.  Conventions follow the reference:
 * elNodes is 1-based [ne][10] in fcVM's internal node order, dofs are 3*(node-1)+k, the
 * stress vector order is (xx, yy, zz, xy, zx, yz), 4 Gauss points per element.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* fcVM.py:587-596 gaussPoints(): 4-point rule for the 10-node tetrahedron */
static const double GP10[4][4] = {
    {0.138196601125011, 0.138196601125011, 0.138196601125011, 0.041666666666667},
    {0.585410196624968, 0.138196601125011, 0.138196601125011, 0.041666666666667},
    {0.138196601125011, 0.585410196624968, 0.138196601125011, 0.041666666666667},
    {0.138196601125011, 0.138196601125011, 0.585410196624968, 0.041666666666667}};
/* fcVM.py:598-609: 6-point rule for the 6-node triangle */
static const double GP6[6][3] = {
    {0.445948490915965, 0.445948490915965, 0.111690794839005},
    {0.10810301816807, 0.445948490915965, 0.111690794839005},
    {0.445948490915965, 0.10810301816807, 0.111690794839005},
    {0.091576213509771, 0.091576213509771, 0.054975871827661},
    {0.816847572980458, 0.091576213509771, 0.054975871827661},
    {0.091576213509771, 0.816847572980458, 0.054975871827661}};
/* fcVM.py:611: 2-point rule for the 3-node line */
static const double GP2[2][2] = {{-0.5773502691896257, 1.0}, {0.5773502691896257, 1.0}};

/* fcVM.py:365-380 shp10tet */
static void shp10(double xi, double et, double ze, double *s) {
    double a = 1.0 - xi - et - ze;
    s[0] = (2.0 * a - 1.0) * a;
    s[1] = xi * (2.0 * xi - 1.0);
    s[2] = et * (2.0 * et - 1.0);
    s[3] = ze * (2.0 * ze - 1.0);
    s[4] = 4.0 * xi * a;
    s[5] = 4.0 * xi * et;
    s[6] = 4.0 * et * a;
    s[7] = 4.0 * ze * a;
    s[8] = 4.0 * xi * ze;
    s[9] = 4.0 * et * ze;
}

/* fcVM.py:384-480 dshp10tet: local derivatives, Jacobian, global derivatives dg[3][10].
 * xl[10][3] nodal coordinates.  Returns the (signed) Jacobian determinant. */
static double dshp10(double xi, double et, double ze, const double xl[10][3], double dg[3][10]) {
    double d[3][10];
    double a4 = 4.0 * (1.0 - xi - et - ze);
    memset(d, 0, sizeof d);
    d[0][0] = 1.0 - a4; d[0][1] = 4.0 * xi - 1.0; d[0][4] = 4.0 * (1.0 - 2.0 * xi - et - ze);
    d[0][5] = 4.0 * et; d[0][6] = -4.0 * et; d[0][7] = -4.0 * ze; d[0][8] = 4.0 * ze;
    d[1][0] = 1.0 - a4; d[1][2] = 4.0 * et - 1.0; d[1][4] = -4.0 * xi; d[1][5] = 4.0 * xi;
    d[1][6] = 4.0 * (1.0 - xi - 2.0 * et - ze); d[1][7] = -4.0 * ze; d[1][9] = 4.0 * ze;
    d[2][0] = 1.0 - a4; d[2][3] = 4.0 * ze - 1.0; d[2][4] = -4.0 * xi; d[2][6] = -4.0 * et;
    d[2][7] = 4.0 * (1.0 - xi - et - 2.0 * ze); d[2][8] = 4.0 * xi; d[2][9] = 4.0 * et;
    double xs[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double t = 0.0;
            for (int k = 0; k < 10; k++) t += xl[k][i] * d[j][k];
            xs[i][j] = t;
        }
    double xsj = xs[0][0] * xs[1][1] * xs[2][2] - xs[0][0] * xs[1][2] * xs[2][1] +
                 xs[0][2] * xs[1][0] * xs[2][1] - xs[0][2] * xs[1][1] * xs[2][0] +
                 xs[0][1] * xs[1][2] * xs[2][0] - xs[0][1] * xs[1][0] * xs[2][2];
    double xi_[3][3];
    xi_[0][0] = (xs[1][1] * xs[2][2] - xs[2][1] * xs[1][2]) / xsj;
    xi_[0][1] = (xs[0][2] * xs[2][1] - xs[0][1] * xs[2][2]) / xsj;
    xi_[0][2] = (xs[0][1] * xs[1][2] - xs[0][2] * xs[1][1]) / xsj;
    xi_[1][0] = (xs[1][2] * xs[2][0] - xs[1][0] * xs[2][2]) / xsj;
    xi_[1][1] = (xs[0][0] * xs[2][2] - xs[0][2] * xs[2][0]) / xsj;
    xi_[1][2] = (xs[1][0] * xs[0][2] - xs[0][0] * xs[1][2]) / xsj;
    xi_[2][0] = (xs[1][0] * xs[2][1] - xs[2][0] * xs[1][1]) / xsj;
    xi_[2][1] = (xs[2][0] * xs[0][1] - xs[0][0] * xs[2][1]) / xsj;
    xi_[2][2] = (xs[0][0] * xs[1][1] - xs[1][0] * xs[0][1]) / xsj;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 10; j++) {
            double t = 0.0;
            for (int k = 0; k < 3; k++) t += xi_[k][i] * d[k][j];
            dg[i][j] = t;
        }
    return xsj;
}

/* strain interpolation matrix, fcVM.py:464-478 */
static void bmat_from_dg(const double dg[3][10], double b[6][30]) {
    memset(b, 0, sizeof(double) * 180);
    for (int i = 0; i < 10; i++) {
        int i3 = 3 * i;
        b[0][i3] = dg[0][i];
        b[1][i3 + 1] = dg[1][i];
        b[2][i3 + 2] = dg[2][i];
        b[3][i3] = dg[1][i]; b[3][i3 + 1] = dg[0][i];
        b[4][i3] = dg[2][i]; b[4][i3 + 2] = dg[0][i];
        b[5][i3 + 1] = dg[2][i]; b[5][i3 + 2] = dg[1][i];
    }
}

/* fcVM.py:570-582 hooke */
static void hooke(double e, double nu, double dm[6][6]) {
    double f = e * (1.0 - nu) / (1.0 + nu) / (1.0 - 2.0 * nu);
    double od = nu / (1.0 - nu), sd = 0.5 * (1.0 - 2.0 * nu) / (1.0 - nu);
    memset(dm, 0, sizeof(double) * 36);
    dm[0][0] = dm[1][1] = dm[2][2] = f;
    dm[3][3] = dm[4][4] = dm[5][5] = sd * f;
    dm[0][1] = dm[0][2] = dm[1][2] = dm[1][0] = dm[2][0] = dm[2][1] = od * f;
}

/* fcVM.py:485-541 shape6tri: shape functions, |normal| and unit normal xp of a 6-node
 * triangle; xl[3][6]. */
static double shape6tri(double xi, double et, const double xl[3][6], double shp[6], double xp[3]) {
    double d[2][6];
    shp[0] = (1.0 - xi - et) * (1.0 - 2.0 * xi - 2.0 * et);
    shp[1] = xi * (2.0 * xi - 1.0);
    shp[2] = et * (2.0 * et - 1.0);
    shp[3] = 4.0 * xi * (1.0 - xi - et);
    shp[4] = 4.0 * xi * et;
    shp[5] = 4.0 * et * (1 - xi - et);
    d[0][0] = -3.0 + 4.0 * et + 4.0 * xi; d[0][1] = -1.0 + 4.0 * xi; d[0][2] = 0.0;
    d[0][3] = -4.0 * (-1.0 + et + 2.0 * xi); d[0][4] = 4.0 * et; d[0][5] = -4.0 * et;
    d[1][0] = -3.0 + 4.0 * et + 4.0 * xi; d[1][1] = 0.0; d[1][2] = -1.0 + 4.0 * et;
    d[1][3] = -4.0 * xi; d[1][4] = 4.0 * xi; d[1][5] = -4.0 * (-1.0 + 2.0 * et + xi);
    double xs[2][3];
    for (int a = 0; a < 2; a++)
        for (int c = 0; c < 3; c++) {
            double t = 0.0;
            for (int k = 0; k < 6; k++) t += d[a][k] * xl[c][k];
            xs[a][c] = t;
        }
    xp[0] = xs[0][1] * xs[1][2] - xs[0][2] * xs[1][1];
    xp[1] = xs[0][2] * xs[1][0] - xs[0][0] * xs[1][2];
    xp[2] = xs[0][0] * xs[1][1] - xs[0][1] * xs[1][0];
    double xsj = sqrt(xp[0] * xp[0] + xp[1] * xp[1] + xp[2] * xp[2]);
    xp[0] /= xsj; xp[1] /= xsj; xp[2] /= xsj;
    return xsj;
}

/* fcVM.py:545-565 shape2lin; xle[3][3] */
static double shape2lin(double xi, const double xle[3][3], double shp[3]) {
    double d[3];
    shp[0] = -0.5 * (1.0 - xi) * xi;
    shp[1] = 0.5 * (1.0 + xi) * xi;
    shp[2] = (1.0 + xi) * (1.0 - xi);
    d[0] = xi - 0.5; d[1] = xi + 0.5; d[2] = -2.0 * xi;
    double dx = 0, dy = 0, dz = 0;
    for (int k = 0; k < 3; k++) { dx += xle[0][k] * d[k]; dy += xle[1][k] * d[k]; dz += xle[2][k] * d[k]; }
    return sqrt(dx * dx + dy * dy + dz * dz);
}

/*
 * External load vector glv (without gravity): pressure faces (normal, follows xp), vertex
 * forces, uniformly distributed face tractions, edge line loads.
 * fcVM.py:649-727 (calcGSM) and, with disp != NULL, the follower-pressure variant of
 * fcVM.py:858-883 (calcTSM: pressure acts on the displaced surface).
 * Arrays hold REAL entries only (the reference's dummy first rows are dropped by the caller).
 */
void orc_external_loads(const double *nocoord, const double *disp,
                        int64_t npf, const int64_t *pfaces, const double *pvals,
                        int64_t nv, const int64_t *vnodes, const double *vloads,
                        int64_t nuf, const int64_t *ufaces, const double *uloads,
                        int64_t nedge, const int64_t *edges, const double *eloads, double *glv) {
    double xlf[3][6], shp[6], xp[3];
    for (int64_t f = 0; f < npf; f++) {
        const int64_t *nd = pfaces + 6 * f;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 6; j++)
                xlf[i][j] = nocoord[3 * (nd[j] - 1) + i] + (disp ? disp[3 * (nd[j] - 1) + i] : 0.0);
        for (int g = 0; g < 6; g++) {
            double xsj = shape6tri(GP6[g][0], GP6[g][1], xlf, shp, xp);
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 3; k++)
                    glv[3 * (nd[i] - 1) + k] += shp[i] * pvals[f] * xp[k] * fabs(xsj) * GP6[g][2];
        }
    }
    for (int64_t v = 0; v < nv; v++)
        for (int k = 0; k < 3; k++) glv[3 * (vnodes[v] - 1) + k] += vloads[3 * v + k];
    for (int64_t f = 0; f < nuf; f++) {
        const int64_t *nd = ufaces + 6 * f;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 6; j++) xlf[i][j] = nocoord[3 * (nd[j] - 1) + i];
        for (int g = 0; g < 6; g++) {
            double xsj = shape6tri(GP6[g][0], GP6[g][1], xlf, shp, xp);
            for (int i = 0; i < 6; i++)
                for (int k = 0; k < 3; k++)
                    glv[3 * (nd[i] - 1) + k] += shp[i] * uloads[3 * f + k] * fabs(xsj) * GP6[g][2];
        }
    }
    for (int64_t e = 0; e < nedge; e++) {
        const int64_t *nd = edges + 3 * e;
        double xle[3][3], s3[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) xle[i][j] = nocoord[3 * (nd[j] - 1) + i];
        for (int g = 0; g < 2; g++) {
            double xsj = shape2lin(GP2[g][0], xle, s3);
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++)
                    glv[3 * (nd[i] - 1) + k] += s3[i] * eloads[3 * e + k] * fabs(xsj) * GP2[g][1];
        }
    }
}

/*
 * Element stiffness, gravity load and lower-triangular COO assembly with the displacement
 * boundary conditions folded in: fcVM.py:729-800 (calcGSM) and, when sig/pgp are given, the
 * material part of calcTSM fcVM.py:940-1050 (D - pmat at plastic Gauss points, coordinates
 * updated by disp).  Geometric stiffness (nstep == 1 buckling branch) is NOT restated here.
 *
 * isfix[dof] / fixval[dof] replace the reference's numba Dict `fix`.
 * row/col/stm must hold 465*ne entries; returns the number used (`pos`).
 * glv (+= gravity), modf (+=), x[4*ne][3] Gauss-point coordinates, *V volume.
 */
int64_t orc_calc_stiffness(int64_t ne, const int64_t *elNodes, const double *nocoord,
                           const double *disp, double E, double nu, double density,
                           const uint8_t *isfix, const double *fixval, double gx, double gy,
                           double gz, const double *sig, const uint8_t *pgp, double Et_E,
                           int64_t *row, int64_t *col, double *stm, double *glv, double *modf,
                           double *x, double *V) {
    double dmat[6][6], dm[6][6], b[6][30], dg[3][10], xl[10][3], shp[10];
    static double esm[30][30];
    double db[6][30];
    hooke(E, nu, dmat);
    double G = E / (1.0 + nu) / 2.0;
    if (Et_E > 0.95) Et_E = 0.95;
    double Et = Et_E * E, H = Et / (1.0 - Et_E);
    int64_t pos = 0;
    double vol = 0.0;
    for (int64_t el = 0; el < ne; el++) {
        const int64_t *nodes = elNodes + 10 * el;
        double gamma[30];
        int64_t dof[30];
        memset(esm, 0, sizeof esm);
        memset(gamma, 0, sizeof gamma);
        for (int j = 0; j < 10; j++)
            for (int i = 0; i < 3; i++)
                xl[j][i] = nocoord[3 * (nodes[j] - 1) + i] + (disp ? disp[3 * (nodes[j] - 1) + i] : 0.0);
        for (int g = 0; g < 4; g++) {
            shp10(GP10[g][0], GP10[g][1], GP10[g][2], shp);
            double xsj = dshp10(GP10[g][0], GP10[g][1], GP10[g][2], xl, dg);
            bmat_from_dg(dg, b);
            memcpy(dm, dmat, sizeof dm);
            if (pgp && pgp[4 * el + g]) { /* fcVM.py:983-998 */
                const double *s0 = sig + 24 * el + 6 * g;
                double s[6] = {s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
                double p = (s[0] + s[1] + s[2]) / 3.0;
                s[0] -= p; s[1] -= p; s[2] -= p;
                double svm = sqrt(1.5 * (s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) +
                                  3.0 * (s[3] * s[3] + s[4] * s[4] + s[5] * s[5]));
                if (svm == 0.0) svm = 1.0;
                double fac = 3.0 * G / (1.0 + H / 3.0 / G) / (svm * svm);
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) dm[i][j] -= fac * s[i] * s[j];
            }
            double w = GP10[g][3] * fabs(xsj);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 30; j++) {
                    double t = 0.0;
                    for (int k = 0; k < 6; k++) t += dm[i][k] * b[k][j];
                    db[i][j] = t;
                }
            for (int i = 0; i < 30; i++)
                for (int j = 0; j < 30; j++) {
                    double t = 0.0;
                    for (int k = 0; k < 6; k++) t += b[k][i] * db[k][j];
                    esm[i][j] += t * w;
                }
            for (int i = 0; i < 10; i++) {
                gamma[3 * i] += gx * density * shp[i] * w;
                gamma[3 * i + 1] += gy * density * shp[i] * w;
                gamma[3 * i + 2] += gz * density * shp[i] * w;
            }
            vol += xsj * GP10[g][3];
            for (int c = 0; c < 3; c++) {
                double t = 0.0;
                for (int k = 0; k < 10; k++) t += xl[k][c] * shp[k];
                x[3 * (4 * el + g) + c] = t;
            }
        }
        for (int i = 0; i < 10; i++) {
            int64_t nd = nodes[i] - 1;
            for (int k = 0; k < 3; k++) {
                glv[3 * nd + k] += gamma[3 * i + k];
                dof[3 * i + k] = 3 * nd + k;
            }
        }
        for (int i = 0; i < 30; i++) { /* fcVM.py:771-796 */
            int64_t di = dof[i];
            if (isfix[di]) {
                row[pos] = di; col[pos] = di; stm[pos] = 1.0;
                modf[di] += fixval[di];
                pos++;
                for (int j = 0; j < i; j++)
                    if (!isfix[dof[j]]) modf[dof[j]] -= esm[i][j] * fixval[di];
            } else {
                for (int j = 0; j <= i; j++) {
                    int64_t dj = dof[j];
                    if (isfix[dj]) {
                        modf[di] -= esm[i][j] * fixval[dj];
                    } else {
                        if (di > dj) { row[pos] = di; col[pos] = dj; }
                        else { row[pos] = dj; col[pos] = di; }
                        stm[pos] = esm[i][j];
                        pos++;
                    }
                }
            }
        }
    }
    *V = vol;
    return pos;
}

/* fcVM.py:2469-2492 vmises_original_optimised: radial return; returns the plastic flag */
static int vmises_return(const double st[6], double sy, double H, double G, double out[6]) {
    double p = (st[0] + st[1] + st[2]) / 3.0;
    double s0 = st[0] - p, s1 = st[1] - p, s2 = st[2] - p;
    double svm = sqrt(1.5 * (s0 * s0 + s1 * s1 + s2 * s2) +
                      3.0 * (st[3] * st[3] + st[4] * st[4] + st[5] * st[5]));
    double fac;
    int pp;
    if (sy > svm) { fac = 1.0; pp = 0; }
    else { fac = (1.0 - (1.0 - sy / svm) * 3.0 * G / (H + 3 * G)); pp = 1; }
    out[0] = fac * s0 + p; out[1] = fac * s1 + p; out[2] = fac * s2 + p;
    out[3] = fac * st[3]; out[4] = fac * st[4]; out[5] = fac * st[5];
    return pp;
}

/*
 * fcVM.py:2197-2464 update_stress_load: strain increment from du, (optionally convected)
 * old stress + D deps = trial stress, radial return, internal force vector qin (+=).
 * LD != 0: updated-Lagrange branch (coordinates + disp_new, stress convection by F).
 */
void orc_update_stress_load(int64_t ne, const int64_t *elNodes, const double *nocoord, double E,
                            double nu, const double *sig_yield, const double *disp_new,
                            const double *du, const double *sig, double *sig_update,
                            double *sig_test_global, double *qin, double Et_E, int LD,
                            uint8_t *pgp) {
    double dmat[6][6], dg[3][10], xl[10][3];
    hooke(E, nu, dmat);
    double G = E / 2.0 / (1 + nu);
    if (Et_E > 0.95) Et_E = 0.95;
    double Et = Et_E * E, H = Et / (1.0 - Et_E);
    for (int64_t el = 0; el < ne; el++) {
        const int64_t *nodes = elNodes + 10 * el;
        double elv[30], dux[10], duy[10], duz[10];
        memset(elv, 0, sizeof elv);
        for (int j = 0; j < 10; j++) {
            int64_t n3 = 3 * (nodes[j] - 1);
            dux[j] = du[n3]; duy[j] = du[n3 + 1]; duz[j] = du[n3 + 2];
            for (int i = 0; i < 3; i++) xl[j][i] = nocoord[n3 + i] + (LD ? disp_new[n3 + i] : 0.0);
        }
        for (int g = 0; g < 4; g++) {
            int64_t ipp = 4 * el + g, ippos = 24 * el + 6 * g;
            double xsj = dshp10(GP10[g][0], GP10[g][1], GP10[g][2], xl, dg);
            double deps[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 10; j++) { /* fcVM.py:2374-2381 */
                deps[0] += dg[0][j] * dux[j];
                deps[1] += dg[1][j] * duy[j];
                deps[2] += dg[2][j] * duz[j];
                deps[3] += dg[1][j] * dux[j] + dg[0][j] * duy[j];
                deps[4] += dg[2][j] * dux[j] + dg[0][j] * duz[j];
                deps[5] += dg[2][j] * duy[j] + dg[1][j] * duz[j];
            }
            double sigc[6];
            if (LD) { /* fcVM.py:2383-2429 */
                const double *s = sig + ippos;
                double st[3][3] = {{s[0], s[3], s[4]}, {s[3], s[1], s[5]}, {s[4], s[5], s[2]}};
                double F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
                for (int i = 0; i < 10; i++)
                    for (int c = 0; c < 3; c++) {
                        F[0][c] += dux[i] * dg[c][i];
                        F[1][c] += duy[i] * dg[c][i];
                        F[2][c] += duz[i] * dg[c][i];
                    }
                double rr = F[0][0] * F[1][1] * F[2][2] - F[0][0] * F[1][2] * F[2][1] +
                            F[0][2] * F[1][0] * F[2][1] - F[0][2] * F[1][1] * F[2][0] +
                            F[0][1] * F[1][2] * F[2][0] - F[0][1] * F[1][0] * F[2][2];
                rr = 1.0 / rr;
                double sc[3][3];
                memset(sc, 0, sizeof sc);
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        for (int k = 0; k < 3; k++)
                            for (int l = 0; l < 3; l++) sc[i][k] += F[i][j] * st[j][l] * F[k][l];
                sigc[0] = rr * sc[0][0]; sigc[1] = rr * sc[1][1]; sigc[2] = rr * sc[2][2];
                sigc[3] = rr * sc[0][1]; sigc[4] = rr * sc[0][2]; sigc[5] = rr * sc[1][2];
            } else {
                for (int j = 0; j < 6; j++) sigc[j] = sig[ippos + j];
            }
            double stt[6], sn[6];
            for (int j = 0; j < 6; j++) {
                double t = sigc[j];
                for (int k = 0; k < 6; k++) t += dmat[j][k] * deps[k];
                stt[j] = t;
                sig_test_global[ippos + j] = t;
            }
            pgp[ipp] = (uint8_t)vmises_return(stt, sig_yield[ipp], H, G, sn);
            for (int j = 0; j < 6; j++) sig_update[ippos + j] = sn[j];
            double w = GP10[g][3] * fabs(xsj);
            for (int j = 0; j < 10; j++) { /* fcVM.py:2449-2454 */
                elv[3 * j] += (dg[0][j] * sn[0] + dg[1][j] * sn[3] + dg[2][j] * sn[4]) * w;
                elv[3 * j + 1] += (dg[1][j] * sn[1] + dg[0][j] * sn[3] + dg[2][j] * sn[5]) * w;
                elv[3 * j + 2] += (dg[2][j] * sn[2] + dg[0][j] * sn[4] + dg[1][j] * sn[5]) * w;
            }
        }
        for (int i = 0; i < 10; i++)
            for (int k = 0; k < 3; k++) qin[3 * (nodes[i] - 1) + k] += elv[3 * i + k];
    }
}

/* fcVM.py:2085-2137 update_PEEQ_CSR (sig_yield, peeq, csr are updated in place) */
void orc_update_peeq_csr(int64_t ne, double E, double nu, const double *sig_test,
                         const double *sig_new, double *sig_yield, double ultimate_strain,
                         double *peeq, double *csr, double *triax, double *pressure,
                         double *sigmises, double *ecr, double Et_E) {
    double G = E / 2.0 / (1 + nu);
    if (Et_E > 0.95) Et_E = 0.95;
    double Et = Et_E * E, H = Et / (1.0 - Et_E);
    if (ultimate_strain == 0.0) ultimate_strain = 1.0e12;
    double alpha = sqrt(2.718281828459045) * ultimate_strain, beta = 1.5;
    for (int64_t ip = 0; ip < 4 * ne; ip++) {
        const double *t = sig_test + 6 * ip, *n = sig_new + 6 * ip;
        double pt = (t[0] + t[1] + t[2]) / 3.0, pn = (n[0] + n[1] + n[2]) / 3.0;
        double t0 = t[0] - pt, t1 = t[1] - pt, t2 = t[2] - pt;
        double n0 = n[0] - pn, n1 = n[1] - pn, n2 = n[2] - pn;
        double svt = sqrt(1.5 * (t0 * t0 + t1 * t1 + t2 * t2) + 3.0 * (t[3] * t[3] + t[4] * t[4] + t[5] * t[5]));
        double svn = sqrt(1.5 * (n0 * n0 + n1 * n1 + n2 * n2) + 3.0 * (n[3] * n[3] + n[4] * n[4] + n[5] * n[5]));
        double DL = 0.0;
        if (svt > sig_yield[ip]) {
            DL = (svt - sig_yield[ip]) / (3.0 * G + H);
            peeq[ip] += DL;
            sig_yield[ip] += Et * DL;
        }
        double T = pn / sig_yield[ip];
        pressure[ip] = pn;
        sigmises[ip] = svn;
        triax[ip] = T;
        double ec = alpha * exp(-beta * T);
        if (ec < 1.0e-6) ec = 1.0e-6;
        ecr[ip] = ec;
        csr[ip] += DL / ec;
    }
}

/* fcVM.py:2497-2554 mapStresses (Gauss-point values are taken as corner-node values) */
void orc_map_stresses(int averaged, int64_t ne, int64_t nn, const int64_t *elNodes,
                      const double *sig, const double *peeq, const double *sigvm,
                      const double *csr, const int16_t *noce, double sig_yield, double *t10s,
                      double *t10peeq, double *t10csr, double *t10svm, double *t10triax) {
    static const int inter[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {1, 3}, {2, 3}};
    memset(t10s, 0, sizeof(double) * 6 * nn);
    memset(t10peeq, 0, sizeof(double) * nn);
    memset(t10csr, 0, sizeof(double) * nn);
    memset(t10svm, 0, sizeof(double) * nn);
    memset(t10triax, 0, sizeof(double) * nn);
    for (int64_t el = 0; el < ne; el++)
        for (int c = 0; c < 4; c++) {
            int64_t nd = elNodes[10 * el + c] - 1;
            for (int k = 0; k < 6; k++) t10s[6 * nd + k] += sig[24 * el + 6 * c + k] / noce[nd];
        }
    for (int64_t el = 0; el < ne; el++)
        for (int c = 0; c < 4; c++) {
            int64_t nd = elNodes[10 * el + c] - 1, ip = 4 * el + c;
            double tr = (sig[6 * ip] + sig[6 * ip + 1] + sig[6 * ip + 2]) / 3.0 / sig_yield;
            if (averaged) {
                t10peeq[nd] += peeq[ip] / noce[nd];
                t10csr[nd] += csr[ip] / noce[nd];
                t10svm[nd] += sigvm[ip] / noce[nd];
                t10triax[nd] += tr / noce[nd];
            } else {
                t10peeq[nd] = fmax(t10peeq[nd], peeq[ip]);
                t10csr[nd] = fmax(t10csr[nd], csr[ip]);
                t10svm[nd] = fmax(t10svm[nd], sigvm[ip]);
                t10triax[nd] = fmax(t10triax[nd], tr);
            }
        }
    for (int64_t el = 0; el < ne; el++) {
        const int64_t *nodes = elNodes + 10 * el;
        for (int m = 0; m < 6; m++) {
            int64_t nd = nodes[4 + m] - 1, a = nodes[inter[m][0]] - 1, b = nodes[inter[m][1]] - 1;
            for (int k = 0; k < 6; k++) t10s[6 * nd + k] = 0.5 * t10s[6 * a + k] + 0.5 * t10s[6 * b + k];
            t10peeq[nd] = 0.5 * t10peeq[a] + 0.5 * t10peeq[b];
            t10csr[nd] = 0.5 * t10csr[a] + 0.5 * t10csr[b];
            t10svm[nd] = 0.5 * t10svm[a] + 0.5 * t10svm[b];
            t10triax[nd] = 0.5 * t10triax[a] + 0.5 * t10triax[b];
        }
    }
}

/*
 * Linear-buckling branch of calcTSM (nstep == 1), fcVM.py:1002-1006, 1051-1062: per element the
 * material stiffness esm (elastic D) and the geometric stiffness
 *   nsm = sum_gp GM^T SM GM |J| w,  GM = kron(dshpg, I3),  SM = kron(sigma, I3)
 * are written as FULL 30x30 COO blocks (no boundary-condition elimination); the material diagonal of
 * fixed dofs is multiplied by 100 (fcVM.py:1060-1061).  row/col/stms/stmg hold 900*ne entries.
 */
void orc_buckling_matrices(int64_t ne, const int64_t *elNodes, const double *nocoord, double E, double nu,
                           const uint8_t *isfix, const double *sig, int64_t *row, int64_t *col,
                           double *stms, double *stmg) {
    double dmat[6][6], b[6][30], dg[3][10], xl[10][3], db[6][30];
    static double esm[30][30], nsm[30][30];
    hooke(E, nu, dmat);
    int64_t pos = 0;
    for (int64_t el = 0; el < ne; el++) {
        const int64_t *nodes = elNodes + 10 * el;
        int64_t dof[30];
        memset(esm, 0, sizeof esm);
        memset(nsm, 0, sizeof nsm);
        for (int j = 0; j < 10; j++)
            for (int i = 0; i < 3; i++) xl[j][i] = nocoord[3 * (nodes[j] - 1) + i];
        for (int g = 0; g < 4; g++) {
            double xsj = dshp10(GP10[g][0], GP10[g][1], GP10[g][2], xl, dg);
            bmat_from_dg(dg, b);
            double w = GP10[g][3] * fabs(xsj);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 30; j++) {
                    double t = 0.0;
                    for (int k = 0; k < 6; k++) t += dmat[i][k] * b[k][j];
                    db[i][j] = t;
                }
            for (int i = 0; i < 30; i++)
                for (int j = 0; j < 30; j++) {
                    double t = 0.0;
                    for (int k = 0; k < 6; k++) t += b[k][i] * db[k][j];
                    esm[i][j] += t * w;
                }
            const double *s0 = sig + 24 * el + 6 * g;
            double s[3][3] = {{s0[0], s0[3], s0[4]}, {s0[3], s0[1], s0[5]}, {s0[4], s0[5], s0[2]}};
            for (int a = 0; a < 10; a++)
                for (int c = 0; c < 10; c++) {
                    double t = 0.0;
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) t += dg[k][a] * s[k][l] * dg[l][c];
                    for (int i = 0; i < 3; i++) nsm[3 * a + i][3 * c + i] += t * w;
                }
        }
        for (int i = 0; i < 10; i++)
            for (int k = 0; k < 3; k++) dof[3 * i + k] = 3 * (nodes[i] - 1) + k;
        for (int i = 0; i < 30; i++)
            for (int j = 0; j < 30; j++) {
                row[pos] = dof[i];
                col[pos] = dof[j];
                stms[pos] = esm[i][j];
                stmg[pos] = nsm[i][j];
                if (i == j && isfix[dof[i]]) stms[pos] *= 100.0;
                pos++;
            }
    }
}
