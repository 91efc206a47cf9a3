/* TEST SCAFFOLDING, declarations only (see fftw3.h next to this file): the four fftw_mpi_* calls of the reference
 * harness's `-a 1` comparison path (run-fft.c:324-334, 424), for a -fsyntax-only check. */
#ifndef OFFT_TEST_SHIM_FFTW3_MPI_H
#define OFFT_TEST_SHIM_FFTW3_MPI_H
#include <stddef.h>
#include <mpi.h>
#include "fftw3.h"
#define FFTW_MPI_TRANSPOSED_OUT (1U << 30)
void fftw_mpi_init(void);
void fftw_mpi_cleanup(void);
fftw_plan fftw_mpi_plan_dft_3d(ptrdiff_t n0, ptrdiff_t n1, ptrdiff_t n2, fftw_complex *in, fftw_complex *out, MPI_Comm comm, int sign, unsigned flags);
fftw_plan fftw_mpi_plan_dft_r2c_3d(ptrdiff_t n0, ptrdiff_t n1, ptrdiff_t n2, double *in, fftw_complex *out, MPI_Comm comm, unsigned flags);
#endif
