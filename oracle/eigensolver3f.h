// eigensolver3f.h — Eigen 3.4.0's EigenSolver<Matrix3f>::compute, eigenvalues() and eigenvectors() restated statement by statement
// for the CPU ORACLE (test infrastructure only; the product never includes this file).
//
// The reference calls it once per Gaussian, Gaussians.h:184-188:
//     EigenSolver<Matrix3f> eigensolver;  eigensolver.compute(io_cov);
//     Vector3f eigenValues = eigensolver.eigenvalues().real();  Matrix3f eigenVectors = eigensolver.eigenvectors().real();
// Eigen is not in this container (README.md:77-79 of the reference names 3.4.0; Poses.h:68 needs 3.4's `reshaped`), so what follows
// is the published algorithm of these files, as recalled, function by function under Eigen's own names:
//     Eigen/src/Eigenvalues/EigenSolver.h            compute, doComputeEigenvectors, eigenvectors
//     Eigen/src/Eigenvalues/RealSchur.h              compute, computeFromHessenberg, computeNormOfT, findSmallSubdiagEntry, splitOffTwoRows,
//                                                    computeShift, initFrancisQRStep, performFrancisQRStep
//     Eigen/src/Eigenvalues/HessenbergDecomposition.h  _compute, matrixH, matrixQ
//     Eigen/src/Householder/Householder.h            makeHouseholder, applyHouseholderOnTheLeft / OnTheRight
//     Eigen/src/Householder/HouseholderSequence.h    evalTo (length 2 <= BlockSize: one reflector after the other on an identity)
//     Eigen/src/Jacobi/Jacobi.h                      makeGivens (real), apply_rotation_in_the_plane
// The code keeps Eigen's loops over a general size n (= 3 here) so that it can be read against those files; nothing is simplified for
// the symmetric input.  What fixes the BITS, beyond the statements themselves:
//   * everything is scalar float on x86-64 without FMA (the reference's -O1, no -march): every multiply, add, divide and sqrt rounds once.
//     Where Eigen evaluates a statement with SSE packets (rotations of columns, products with a contiguous left factor) the packet code
//     performs the same operations per coefficient (pmadd = pmul + padd without FMA; a + b commutes), so the scalar reading is exact;
//   * inner products: the reflectors' `essential.adjoint() * bottom` / `right * essential` have depth <= 2 (order-free); the back
//     substitution's `row.segment(l, n - l + 1).dot(col.segment(...))` is a dynamic-size redux without packet access (a row of a
//     column-major matrix): sequential from the first term; the back transformation `m_eivec.leftCols(j + 1) * m_matT.col(j).segment(0, j + 1)`
//     is a coefficient-based lazy product (3 rows, <= 3 deep, 1 column) whose coefficient is `(lhs.row(i).transpose().cwiseProduct(rhs)).sum()`
//     over a dynamic size without packet access: sequential, (x0 + x1) + x2 -- the one 3-term sum of the solver (EIG_BACK_HALVES is the
//     alternative reading x0 + (x1 + x2));
//   * `bottom -= tau * essential * tmp` multiplies (tau * essential_i) * tmp_j, `right -= tau * tmp * essential.adjoint()` multiplies
//     (tau * tmp_i) * essential_j: C++ associates left to right and the scalar multiple is evaluated first;
//   * squaredNorm() of a fixed 3-vector (eigenvectors()'s normalize) is Eigen's unrolled redux x0 + (x1 + x2);
//   * normalize() then runs `derived() /= numext::sqrt(z)` on a COMPLEX column: DenseBase::operator/=(const Scalar&) takes the real norm as
//     std::complex<float>(nrm, 0) and assigns with div_assign_op over a fixed 3-vector of complex<float>.  packet_traits<complex<float>> has
//     HasDiv, a column of a column-major Matrix3cf has PacketAccessBit and LinearAccessBit, EIGEN_UNALIGNED_VECTORIZE is on: the traversal is
//     LinearVectorizedTraversal with complete unrolling -- ONE Packet2cf (rows 0 and 1) and one scalar (row 2).  SSE's pdiv<Packet2cf> is
//     pmul(a, pconj(b)) / (b.re^2 + b.im^2) per component, so rows 0 and 1 get re' = (re * nrm + 0) / (nrm * nrm + 0); row 2 is
//     std::complex's operator/= = libgcc's __divsc3 = re / nrm for a zero imaginary divisor.  Two DIFFERENT roundings inside one column
//     (EIG_NORMALIZE_SCALAR is the other reading: every row re / nrm);
//   * a complex-conjugate pair (possible only when a trailing 2 x 2 block is pure rounding noise) divides std::complex<float> values:
//     libgcc's __divsc3.  GCC 9 (Ubuntu 20.04, the reference's platform) implements Smith's algorithm in float, stated below without
//     its NaN-recovery tail; GCC >= 11 widens to double instead.  `complex_pairs` counts how often a problem gets there (tests: never).
// Not restated: the branch `m_info != Success` (NoConvergence after 40 * 3 iterations leaves EigenSolver's members uninitialised in the
// reference); `info` reports it and the fit then keeps the input covariance's diagonal -- unreachable for symmetric input (tests).
#ifndef ORACLE_EIGENSOLVER3F_H
#define ORACLE_EIGENSOLVER3F_H

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace eigen34 {

typedef float Scalar;
enum { N = 3 };

struct Matrix3f {
    Scalar a[N][N];  // a[row][col]
    Scalar& operator()(int r, int c) { return a[r][c]; }
    Scalar operator()(int r, int c) const { return a[r][c]; }
};

struct EigenSolver3f {
    Scalar eivalues_re[N], eivalues_im[N];  // m_eivalues
    Matrix3f eivec;                          // m_eivec: the pseudo-eigenvectors (real / imaginary parts in adjacent columns for a pair)
    Matrix3f matT;                           // m_matT
    Matrix3f V_re;                           // eigenvectors().real()
    int info = 0;                            // 0 Success, 1 NumericalIssue, 2 NoConvergence
    int iterations = 0;                      // totalIter of RealSchur::computeFromHessenberg
    int complex_pairs = 0;                   // 2 x 2 blocks left on the diagonal of T
};

// ---- Householder.h ---------------------------------------------------------------------------------------------------------
// MatrixBase::makeHouseholder(essential, tau, beta) on a vector of `n` coefficients v[0 .. n-1]; essential has n - 1 coefficients.
static inline void makeHouseholder(const Scalar* v, int n, Scalar* essential, Scalar& tau, Scalar& beta) {
    Scalar tailSqNorm = Scalar(0);  // size() == 1 ? 0 : tail.squaredNorm(); at most two terms here
    for (int k = 1; k < n; ++k) tailSqNorm = k == 1 ? v[k] * v[k] : tailSqNorm + v[k] * v[k];
    const Scalar c0 = v[0];
    const Scalar tol = FLT_MIN;
    if (tailSqNorm <= tol) {  // (the imaginary part of a real c0 is zero)
        tau = Scalar(0);
        beta = c0;
        for (int k = 0; k < n - 1; ++k) essential[k] = Scalar(0);
    } else {
        beta = std::sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= Scalar(0)) beta = -beta;
        for (int k = 0; k < n - 1; ++k) essential[k] = v[k + 1] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
}
// M.block(r0, c0, rows, cols).applyHouseholderOnTheLeft(essential, tau, workspace)
static inline void applyHouseholderOnTheLeft(Matrix3f& M, int r0, int c0, int rows, int cols, const Scalar* essential, Scalar tau) {
    if (rows == 1) {
        const Scalar f = Scalar(1) - tau;  // *this *= Scalar(1) - tau
        for (int j = 0; j < cols; ++j) M(r0, c0 + j) = M(r0, c0 + j) * f;
    } else if (tau != Scalar(0)) {
        Scalar tmp[N];
        for (int j = 0; j < cols; ++j) {  // tmp.noalias() = essential.adjoint() * bottom
            Scalar s = essential[0] * M(r0 + 1, c0 + j);
            for (int k = 1; k < rows - 1; ++k) s = s + essential[k] * M(r0 + 1 + k, c0 + j);
            tmp[j] = s;
        }
        for (int j = 0; j < cols; ++j) tmp[j] = tmp[j] + M(r0, c0 + j);               // tmp += this->row(0)
        for (int j = 0; j < cols; ++j) M(r0, c0 + j) = M(r0, c0 + j) - tau * tmp[j];  // this->row(0) -= tau * tmp
        for (int i = 0; i < rows - 1; ++i)                                            // bottom.noalias() -= tau * essential * tmp
            for (int j = 0; j < cols; ++j) M(r0 + 1 + i, c0 + j) = M(r0 + 1 + i, c0 + j) - (tau * essential[i]) * tmp[j];
    }
}
// M.block(r0, c0, rows, cols).applyHouseholderOnTheRight(essential, tau, workspace)
static inline void applyHouseholderOnTheRight(Matrix3f& M, int r0, int c0, int rows, int cols, const Scalar* essential, Scalar tau) {
    if (cols == 1) {
        const Scalar f = Scalar(1) - tau;
        for (int i = 0; i < rows; ++i) M(r0 + i, c0) = M(r0 + i, c0) * f;
    } else if (tau != Scalar(0)) {
        Scalar tmp[N];
        for (int i = 0; i < rows; ++i) {  // tmp.noalias() = right * essential
            Scalar s = M(r0 + i, c0 + 1) * essential[0];
            for (int k = 1; k < cols - 1; ++k) s = s + M(r0 + i, c0 + 1 + k) * essential[k];
            tmp[i] = s;
        }
        for (int i = 0; i < rows; ++i) tmp[i] = tmp[i] + M(r0 + i, c0);               // tmp += this->col(0)
        for (int i = 0; i < rows; ++i) M(r0 + i, c0) = M(r0 + i, c0) - tau * tmp[i];  // this->col(0) -= tau * tmp
        for (int i = 0; i < rows; ++i)                                                // right.noalias() -= tau * tmp * essential.adjoint()
            for (int j = 0; j < cols - 1; ++j) M(r0 + i, c0 + 1 + j) = M(r0 + i, c0 + 1 + j) - (tau * tmp[i]) * essential[j];
    }
}

// ---- Jacobi.h --------------------------------------------------------------------------------------------------------------
struct JacobiRotation {
    Scalar c, s;
    JacobiRotation adjoint() const { return JacobiRotation{c, -s}; }    // (conj(c), -s)
    JacobiRotation transpose() const { return JacobiRotation{c, -s}; }  // (c, -conj(s))
};
// JacobiRotation::makeGivens(p, q), the specialisation for reals
static inline JacobiRotation makeGivens(Scalar p, Scalar q) {
    JacobiRotation r;
    if (q == Scalar(0)) {
        r.c = p < Scalar(0) ? Scalar(-1) : Scalar(1);
        r.s = Scalar(0);
    } else if (p == Scalar(0)) {
        r.c = Scalar(0);
        r.s = q < Scalar(0) ? Scalar(1) : Scalar(-1);
    } else if (std::fabs(p) > std::fabs(q)) {
        const Scalar t = q / p;
        Scalar u = std::sqrt(Scalar(1) + t * t);
        if (p < Scalar(0)) u = -u;
        r.c = Scalar(1) / u;
        r.s = -t * r.c;
    } else {
        const Scalar t = p / q;
        Scalar u = std::sqrt(Scalar(1) + t * t);
        if (q < Scalar(0)) u = -u;
        r.s = -Scalar(1) / u;
        r.c = -t * r.s;
    }
    return r;
}
// internal::apply_rotation_in_the_plane(x, y, j): x_i <- c x_i + s y_i, y_i <- -s x_i + c y_i
static inline void apply_rotation_in_the_plane(Scalar& x, Scalar& y, const JacobiRotation& j) {
    const Scalar xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}
// M.block(.., c0, .., cols).applyOnTheLeft(p, q, j): rows p and q of the block's columns
static inline void applyOnTheLeft(Matrix3f& M, int c0, int cols, int p, int q, const JacobiRotation& j) {
    if (j.c == Scalar(1) && j.s == Scalar(0)) return;
    for (int k = 0; k < cols; ++k) apply_rotation_in_the_plane(M(p, c0 + k), M(q, c0 + k), j);
}
// M.topRows(rows).applyOnTheRight(p, q, j): columns p and q, with j.transpose()
static inline void applyOnTheRight(Matrix3f& M, int rows, int p, int q, const JacobiRotation& j) {
    const JacobiRotation jt = j.transpose();
    if (jt.c == Scalar(1) && jt.s == Scalar(0)) return;
    for (int k = 0; k < rows; ++k) apply_rotation_in_the_plane(M(k, p), M(k, q), jt);
}

// ---- HessenbergDecomposition.h -----------------------------------------------------------------------------------------------
// _compute(matA, hCoeffs, temp): the packed result (reflectors below the subdiagonal) and the Householder coefficients
static inline void hessenberg_compute(Matrix3f& matA, Scalar* hCoeffs) {
    const int n = N;
    for (int i = 0; i < n - 1; ++i) {
        const int remainingSize = n - i - 1;
        Scalar beta, h;
        // matA.col(i).tail(remainingSize).makeHouseholderInPlace(h, beta): the essential part overwrites the tail's tail
        Scalar v[N], ess[N];
        for (int k = 0; k < remainingSize; ++k) v[k] = matA(i + 1 + k, i);
        makeHouseholder(v, remainingSize, ess, h, beta);
        for (int k = 0; k < remainingSize - 1; ++k) matA(i + 2 + k, i) = ess[k];
        matA(i + 1, i) = beta;
        hCoeffs[i] = h;
        // A = H A
        applyHouseholderOnTheLeft(matA, i + 1, i + 1, remainingSize, remainingSize, ess, h);
        // A = A H'
        applyHouseholderOnTheRight(matA, 0, i + 1, n, remainingSize, ess, h /* numext::conj(h) */);
    }
}

// ---- RealSchur.h -----------------------------------------------------------------------------------------------------------
struct RealSchur3f {
    Matrix3f matT, matU;
    int info = 0, totalIter = 0;

    Scalar computeNormOfT() const {
        Scalar norm(0);
        for (int j = 0; j < N; ++j) {  // norm += m_matT.col(j).segment(0, min(size, j + 2)).cwiseAbs().sum(): fewer than four terms, sequential
            const int len = std::min<int>(N, j + 2);
            Scalar s = std::fabs(matT(0, j));
            for (int i = 1; i < len; ++i) s = s + std::fabs(matT(i, j));
            norm = norm + s;
        }
        return norm;
    }
    int findSmallSubdiagEntry(int iu, Scalar considerAsZero) const {
        int res = iu;
        while (res > 0) {
            Scalar s = std::fabs(matT(res - 1, res - 1)) + std::fabs(matT(res, res));
            s = std::max(s * FLT_EPSILON, considerAsZero);
            if (std::fabs(matT(res, res - 1)) <= s) break;
            res--;
        }
        return res;
    }
    void splitOffTwoRows(int iu, Scalar exshift) {
        const int size = N;
        const Scalar p = Scalar(0.5) * (matT(iu - 1, iu - 1) - matT(iu, iu));
        const Scalar q = p * p + matT(iu, iu - 1) * matT(iu - 1, iu);
        matT(iu, iu) = matT(iu, iu) + exshift;
        matT(iu - 1, iu - 1) = matT(iu - 1, iu - 1) + exshift;
        if (q >= Scalar(0)) {  // two real eigenvalues
            const Scalar z = std::sqrt(std::fabs(q));
            const JacobiRotation rot = p >= Scalar(0) ? makeGivens(p + z, matT(iu, iu - 1)) : makeGivens(p - z, matT(iu, iu - 1));
            applyOnTheLeft(matT, iu - 1, size - iu + 1, iu - 1, iu, rot.adjoint());  // m_matT.rightCols(size - iu + 1).applyOnTheLeft(iu - 1, iu, rot.adjoint())
            applyOnTheRight(matT, iu + 1, iu - 1, iu, rot);                          // m_matT.topRows(iu + 1).applyOnTheRight(iu - 1, iu, rot)
            matT(iu, iu - 1) = Scalar(0);
            applyOnTheRight(matU, size, iu - 1, iu, rot);  // m_matU.applyOnTheRight(iu - 1, iu, rot)
        }
        if (iu > 1) matT(iu - 1, iu - 2) = Scalar(0);
    }
    void computeShift(int iu, int iter, Scalar& exshift, Scalar* shiftInfo) {
        shiftInfo[0] = matT(iu, iu);
        shiftInfo[1] = matT(iu - 1, iu - 1);
        shiftInfo[2] = matT(iu, iu - 1) * matT(iu - 1, iu);
        if (iter == 10) {  // Wilkinson's original ad hoc shift
            exshift = exshift + shiftInfo[0];
            for (int i = 0; i <= iu; ++i) matT(i, i) = matT(i, i) - shiftInfo[0];
            const Scalar s = std::fabs(matT(iu, iu - 1)) + std::fabs(matT(iu - 1, iu - 2));
            shiftInfo[0] = Scalar(0.75) * s;
            shiftInfo[1] = Scalar(0.75) * s;
            shiftInfo[2] = Scalar(-0.4375) * s * s;
        }
        if (iter == 30) {  // MATLAB's new ad hoc shift
            Scalar s = (shiftInfo[1] - shiftInfo[0]) / Scalar(2.0);
            s = s * s + shiftInfo[2];
            if (s > Scalar(0)) {
                s = std::sqrt(s);
                if (shiftInfo[1] < shiftInfo[0]) s = -s;
                s = s + (shiftInfo[1] - shiftInfo[0]) / Scalar(2.0);
                s = shiftInfo[0] - shiftInfo[2] / s;
                exshift = exshift + s;
                for (int i = 0; i <= iu; ++i) matT(i, i) = matT(i, i) - s;
                shiftInfo[0] = shiftInfo[1] = shiftInfo[2] = Scalar(0.964);
            }
        }
    }
    void initFrancisQRStep(int il, int iu, const Scalar* shiftInfo, int& im, Scalar* v) const {
        for (im = iu - 2; im >= il; --im) {
            const Scalar Tmm = matT(im, im);
            const Scalar r = shiftInfo[0] - Tmm;
            const Scalar s = shiftInfo[1] - Tmm;
            v[0] = (r * s - shiftInfo[2]) / matT(im + 1, im) + matT(im, im + 1);
            v[1] = matT(im + 1, im + 1) - Tmm - r - s;
            v[2] = matT(im + 2, im + 1);
            if (im == il) break;
            const Scalar lhs = matT(im, im - 1) * (std::fabs(v[1]) + std::fabs(v[2]));
            const Scalar rhs = v[0] * (std::fabs(matT(im - 1, im - 1)) + std::fabs(Tmm) + std::fabs(matT(im + 1, im + 1)));
            if (std::fabs(lhs) < FLT_EPSILON * rhs) break;
        }
    }
    void performFrancisQRStep(int il, int im, int iu, const Scalar* firstHouseholderVector) {
        const int size = N;
        for (int k = im; k <= iu - 2; ++k) {
            const bool firstIteration = (k == im);
            Scalar v[3];
            if (firstIteration)
                v[0] = firstHouseholderVector[0], v[1] = firstHouseholderVector[1], v[2] = firstHouseholderVector[2];
            else
                v[0] = matT(k, k - 1), v[1] = matT(k + 1, k - 1), v[2] = matT(k + 2, k - 1);
            Scalar tau, beta, ess[2];
            makeHouseholder(v, 3, ess, tau, beta);
            if (beta != Scalar(0)) {
                if (firstIteration && k > il)
                    matT(k, k - 1) = -matT(k, k - 1);
                else if (!firstIteration)
                    matT(k, k - 1) = beta;
                applyHouseholderOnTheLeft(matT, k, k, 3, size - k, ess, tau);
                applyHouseholderOnTheRight(matT, 0, k, std::min(iu, k + 3) + 1, 3, ess, tau);
                applyHouseholderOnTheRight(matU, 0, k, size, 3, ess, tau);
            }
        }
        Scalar v[2] = {matT(iu - 1, iu - 2), matT(iu, iu - 2)};
        Scalar tau, beta, ess[1];
        makeHouseholder(v, 2, ess, tau, beta);
        if (beta != Scalar(0)) {
            matT(iu - 1, iu - 2) = beta;
            applyHouseholderOnTheLeft(matT, iu - 1, iu - 1, 2, size - iu + 1, ess, tau);
            applyHouseholderOnTheRight(matT, 0, iu - 1, iu + 1, 2, ess, tau);
            applyHouseholderOnTheRight(matU, 0, iu - 1, size, 2, ess, tau);
        }
        for (int i = im + 2; i <= iu; ++i) {  // clean up pollution due to round-off errors
            matT(i, i - 2) = Scalar(0);
            if (i > im + 2) matT(i, i - 3) = Scalar(0);
        }
    }
    void computeFromHessenberg() {
        const int maxIters = 40 * N;  // m_maxIterationsPerRow * matrixH.rows()
        int iu = N - 1;
        int iter = 0;
        totalIter = 0;
        Scalar exshift(0);
        const Scalar norm = computeNormOfT();
        const Scalar considerAsZero = std::max<Scalar>(norm * (FLT_EPSILON * FLT_EPSILON), FLT_MIN);
        if (norm != Scalar(0)) {
            while (iu >= 0) {
                const int il = findSmallSubdiagEntry(iu, considerAsZero);
                if (il == iu) {  // one root found
                    matT(iu, iu) = matT(iu, iu) + exshift;
                    if (iu > 0) matT(iu, iu - 1) = Scalar(0);
                    iu--;
                    iter = 0;
                } else if (il == iu - 1) {  // two roots found
                    splitOffTwoRows(iu, exshift);
                    iu -= 2;
                    iter = 0;
                } else {  // no convergence yet
                    Scalar firstHouseholderVector[3] = {0, 0, 0}, shiftInfo[3];
                    computeShift(iu, iter, exshift, shiftInfo);
                    iter = iter + 1;
                    totalIter = totalIter + 1;
                    if (totalIter > maxIters) break;
                    int im;
                    initFrancisQRStep(il, iu, shiftInfo, im, firstHouseholderVector);
                    performFrancisQRStep(il, im, iu, firstHouseholderVector);
                }
            }
        }
        info = totalIter <= maxIters ? 0 : 2;
    }
    // RealSchur::compute(matrix, computeU = true)
    void compute(const Matrix3f& matrix) {
        Scalar scale = Scalar(0);  // matrix.cwiseAbs().maxCoeff()
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) scale = std::max(scale, std::fabs(matrix(i, j)));
        if (scale < FLT_MIN) {
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) matT(i, j) = Scalar(0), matU(i, j) = i == j ? Scalar(1) : Scalar(0);
            info = 0;
            return;
        }
        // Step 1: m_hess.compute(matrix / scale)
        Matrix3f packed;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) packed(i, j) = matrix(i, j) / scale;
        Scalar hCoeffs[N - 1];
        hessenberg_compute(packed, hCoeffs);
        // Step 2: m_hess.matrixQ().evalTo(m_matU, workspace): HouseholderSequence(packed, hCoeffs.conjugate()), length n - 1, shift 1
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) matU(i, j) = i == j ? Scalar(1) : Scalar(0);
        for (int k = N - 2; k >= 0; --k) {
            const int cornerSize = N - k - 1;
            Scalar ess[N];
            for (int r = 0; r < cornerSize - 1; ++r) ess[r] = packed(k + 2 + r, k);  // essentialVector(k): column k below row k + shift
            applyHouseholderOnTheLeft(matU, N - cornerSize, N - cornerSize, cornerSize, cornerSize, ess, hCoeffs[k]);
        }
        // matrixH(): the packed matrix with everything below the subdiagonal cleared
        matT = packed;
        for (int i = 2; i < N; ++i)
            for (int j = 0; j + 2 <= i; ++j) matT(i, j) = Scalar(0);
        computeFromHessenberg();
        for (int j = 0; j < N; ++j)  // m_matT *= scale
            for (int i = 0; i < N; ++i) matT(i, j) = matT(i, j) * scale;
    }
};

// std::complex<float> division as libgcc 9's __divsc3 computes it (Smith's algorithm in float, without the NaN-recovery tail)
static inline void complex_div(Scalar a, Scalar b, Scalar c, Scalar d, Scalar& x, Scalar& y) {
    if (std::fabs(c) < std::fabs(d)) {
        const Scalar ratio = c / d;
        const Scalar denom = (c * ratio) + d;
        x = ((a * ratio) + b) / denom;
        y = ((b * ratio) - a) / denom;
    } else {
        const Scalar ratio = d / c;
        const Scalar denom = (d * ratio) + c;
        x = ((b * ratio) + a) / denom;
        y = (b - (a * ratio)) / denom;
    }
}

// ---- EigenSolver.h ---------------------------------------------------------------------------------------------------------
// real part of (re, im) / (nrm, 0) as row `row` of matV.col(j).normalize() computes it (see the header): rows 0 and 1 through SSE's
// pdiv<Packet2cf> -- pmul(a, pconj(b)) gives re * nrm + (-(im * -0)), the divisor is nrm * nrm + 0 * 0 --, row 2 through __divsc3 (Smith:
// ratio = 0 / nrm, denom = 0 * ratio + nrm, x = (im * ratio + re) / denom)
static inline Scalar normalized_real_part(Scalar re, Scalar im, Scalar nrm, int row) {
#ifndef ORC_VAR_EIG_NORMALIZE_SCALAR
    if (row < 2) return (re * nrm + (-(im * Scalar(-0.0f)))) / (nrm * nrm + Scalar(0) * Scalar(0));
#endif
    const Scalar ratio = Scalar(0) / nrm;
    const Scalar denom = Scalar(0) * ratio + nrm;
    return (im * ratio + re) / denom;
}
// row(i).segment(l, len).dot(col(n).segment(l, len)): sequential (see the header)
static inline Scalar row_dot_col(const Matrix3f& T, int i, int n, int l, int len) {
    Scalar r = T(i, l) * T(l, n);
    for (int k = 1; k < len; ++k) r = r + T(i, l + k) * T(l + k, n);
    return r;
}
static inline void doComputeEigenvectors(EigenSolver3f& es) {
    Matrix3f& m_matT = es.matT;
    const int size = N;
    const Scalar eps = FLT_EPSILON;
    Scalar norm(0);
    for (int j = 0; j < size; ++j) {  // norm += m_matT.row(j).segment(max(j - 1, 0), size - max(j - 1, 0)).cwiseAbs().sum()
        const int first = std::max(j - 1, 0);
        Scalar s = std::fabs(m_matT(j, first));
        for (int k = first + 1; k < size; ++k) s = s + std::fabs(m_matT(j, k));
        norm = norm + s;
    }
    if (norm == Scalar(0)) return;
    for (int n = size - 1; n >= 0; n--) {
        const Scalar p = es.eivalues_re[n];
        const Scalar q = es.eivalues_im[n];
        if (q == Scalar(0)) {  // Scalar vector
            Scalar lastr(0), lastw(0);
            int l = n;
            m_matT(n, n) = Scalar(1);
            for (int i = n - 1; i >= 0; i--) {
                const Scalar w = m_matT(i, i) - p;
                const Scalar r = row_dot_col(m_matT, i, n, l, n - l + 1);
                if (es.eivalues_im[i] < Scalar(0)) {
                    lastw = w;
                    lastr = r;
                } else {
                    l = i;
                    if (es.eivalues_im[i] == Scalar(0)) {
                        if (w != Scalar(0))
                            m_matT(i, n) = -r / w;
                        else
                            m_matT(i, n) = -r / (eps * norm);
                    } else {  // Solve real equations
                        const Scalar x = m_matT(i, i + 1);
                        const Scalar y = m_matT(i + 1, i);
                        const Scalar denom = (es.eivalues_re[i] - p) * (es.eivalues_re[i] - p) + es.eivalues_im[i] * es.eivalues_im[i];
                        const Scalar t = (x * lastr - lastw * r) / denom;
                        m_matT(i, n) = t;
                        if (std::fabs(x) > std::fabs(lastw))
                            m_matT(i + 1, n) = (-r - w * t) / x;
                        else
                            m_matT(i + 1, n) = (-lastr - y * t) / lastw;
                    }
                    const Scalar t = std::fabs(m_matT(i, n));  // Overflow control
                    if ((eps * t) * t > Scalar(1))
                        for (int k = i; k < size; ++k) m_matT(k, n) = m_matT(k, n) / t;  // m_matT.col(n).tail(size - i) /= t
                }
            }
        } else if (q < Scalar(0) && n > 0) {  // Complex vector
            Scalar lastra(0), lastsa(0), lastw(0);
            int l = n - 1;
            if (std::fabs(m_matT(n, n - 1)) > std::fabs(m_matT(n - 1, n))) {
                m_matT(n - 1, n - 1) = q / m_matT(n, n - 1);
                m_matT(n - 1, n) = -(m_matT(n, n) - p) / m_matT(n, n - 1);
            } else {
                Scalar cr, ci;  // ComplexScalar(0, -m_matT(n-1, n)) / ComplexScalar(m_matT(n-1, n-1) - p, q)
                complex_div(Scalar(0), -m_matT(n - 1, n), m_matT(n - 1, n - 1) - p, q, cr, ci);
                m_matT(n - 1, n - 1) = cr;
                m_matT(n - 1, n) = ci;
            }
            m_matT(n, n - 1) = Scalar(0);
            m_matT(n, n) = Scalar(1);
            for (int i = n - 2; i >= 0; i--) {
                const Scalar ra = row_dot_col(m_matT, i, n - 1, l, n - l + 1);
                const Scalar sa = row_dot_col(m_matT, i, n, l, n - l + 1);
                const Scalar w = m_matT(i, i) - p;
                if (es.eivalues_im[i] < Scalar(0)) {
                    lastw = w;
                    lastra = ra;
                    lastsa = sa;
                } else {
                    l = i;
                    if (es.eivalues_im[i] == Scalar(0)) {
                        Scalar cr, ci;  // ComplexScalar(-ra, -sa) / ComplexScalar(w, q)
                        complex_div(-ra, -sa, w, q, cr, ci);
                        m_matT(i, n - 1) = cr;
                        m_matT(i, n) = ci;
                    } else {
                        // "Solve complex equations": needs two 2 x 2 blocks, impossible for size 3
                    }
                    const Scalar t = std::max(std::fabs(m_matT(i, n - 1)), std::fabs(m_matT(i, n)));  // Overflow control
                    if ((eps * t) * t > Scalar(1))
                        for (int c = n - 1; c <= n; ++c)  // m_matT.block(i, n - 1, size - i, 2) /= t
                            for (int k = i; k < size; ++k) m_matT(k, c) = m_matT(k, c) / t;
                }
            }
            (void)lastra, (void)lastsa, (void)lastw;
            n--;  // a pair of complex conjugate eigenvalues: skip them both
        }
    }
    // Back transformation to get eigenvectors of original matrix
    for (int j = size - 1; j >= 0; j--) {
        Scalar m_tmp[N];  // m_tmp.noalias() = m_eivec.leftCols(j + 1) * m_matT.col(j).segment(0, j + 1)
        for (int i = 0; i < size; ++i) {
#ifdef ORC_VAR_EIG_BACK_HALVES
            if (j == 2) {
                m_tmp[i] = es.eivec(i, 0) * m_matT(0, j) + (es.eivec(i, 1) * m_matT(1, j) + es.eivec(i, 2) * m_matT(2, j));
                continue;
            }
#endif
            Scalar s = es.eivec(i, 0) * m_matT(0, j);
            for (int k = 1; k <= j; ++k) s = s + es.eivec(i, k) * m_matT(k, j);
            m_tmp[i] = s;
        }
        for (int i = 0; i < size; ++i) es.eivec(i, j) = m_tmp[i];  // m_eivec.col(j) = m_tmp
    }
}

// EigenSolver<Matrix3f>::compute(matrix, computeEigenvectors = true), then eigenvalues().real() and eigenvectors().real()
static inline void eigensolver_compute(const Matrix3f& matrix, EigenSolver3f& es) {
    RealSchur3f schur;
    schur.compute(matrix);
    es.info = schur.info;
    es.iterations = schur.totalIter;
    es.complex_pairs = 0;
    for (int i = 0; i < N; ++i) es.eivalues_re[i] = es.eivalues_im[i] = Scalar(0);
    es.matT = schur.matT;
    es.eivec = schur.matU;
    es.V_re = schur.matU;
    if (es.info != 0) return;
    Matrix3f& m_matT = es.matT;
    int i = 0;
    while (i < N) {  // Compute eigenvalues from matT
        if (i == N - 1 || m_matT(i + 1, i) == Scalar(0)) {
            es.eivalues_re[i] = m_matT(i, i);
            es.eivalues_im[i] = Scalar(0);
            if (!std::isfinite(es.eivalues_re[i])) {
                es.info = 1;
                return;
            }
            ++i;
        } else {
            const Scalar p = Scalar(0.5) * (m_matT(i, i) - m_matT(i + 1, i + 1));
            Scalar z;
            {  // z = sqrt(abs(p * p + m_matT(i+1, i) * m_matT(i, i+1))) without overflow
                Scalar t0 = m_matT(i + 1, i);
                Scalar t1 = m_matT(i, i + 1);
                const Scalar maxval = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
                t0 = t0 / maxval;
                t1 = t1 / maxval;
                const Scalar p0 = p / maxval;
                z = maxval * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
            }
            es.eivalues_re[i] = m_matT(i + 1, i + 1) + p, es.eivalues_im[i] = z;
            es.eivalues_re[i + 1] = m_matT(i + 1, i + 1) + p, es.eivalues_im[i + 1] = -z;
            es.complex_pairs++;
            if (!(std::isfinite(es.eivalues_re[i]) && std::isfinite(z))) {
                es.info = 1;
                return;
            }
            i += 2;
        }
    }
    doComputeEigenvectors(es);
    // eigenvectors(): normalised columns; .real() keeps the real parts
    const Scalar precision = Scalar(2) * FLT_EPSILON;
    for (int j = 0; j < N; ++j) {
        // internal::isMuchSmallerThan(imag, real, precision): abs(imag) <= abs(real) * precision
        if (std::fabs(es.eivalues_im[j]) <= std::fabs(es.eivalues_re[j]) * precision || j + 1 == N) {
            // matV.col(j) = m_eivec.col(j).cast<ComplexScalar>(); matV.col(j).normalize()
            Scalar x[N];
            for (int r = 0; r < N; ++r) x[r] = es.eivec(r, j) * es.eivec(r, j) + Scalar(0) * Scalar(0);  // abs2 of (re, 0)
            const Scalar z = x[0] + (x[1] + x[2]);
            for (int r = 0; r < N; ++r) es.V_re(r, j) = es.eivec(r, j);
            if (z > Scalar(0)) {
                const Scalar nrm = std::sqrt(z);
                for (int r = 0; r < N; ++r) es.V_re(r, j) = normalized_real_part(es.eivec(r, j), Scalar(0), nrm, r);
            }
        } else {
            // a pair: matV(i, j) = (eivec(i, j), eivec(i, j+1)), matV(i, j+1) = its conjugate; both normalised; the real parts coincide
            Scalar x[N];
            for (int r = 0; r < N; ++r) x[r] = es.eivec(r, j) * es.eivec(r, j) + es.eivec(r, j + 1) * es.eivec(r, j + 1);
            const Scalar z = x[0] + (x[1] + x[2]);
            for (int c = j; c <= j + 1; ++c)
                for (int r = 0; r < N; ++r) es.V_re(r, c) = es.eivec(r, j);
            if (z > Scalar(0)) {
                const Scalar nrm = std::sqrt(z);
                for (int r = 0; r < N; ++r) {  // the conjugate column divides (re, -im): the same real part up to the sign of a zero
                    es.V_re(r, j) = normalized_real_part(es.eivec(r, j), es.eivec(r, j + 1), nrm, r);
                    es.V_re(r, j + 1) = normalized_real_part(es.eivec(r, j), -es.eivec(r, j + 1), nrm, r);
                }
            }
            ++j;
        }
    }
}

}  // namespace eigen34
#endif
