/* TEST SCAFFOLDING, declarations only -- not FFTW, not a reference build.
 * tests/test_boundary_pin.py runs `gcc -fsyntax-only` on the reference's own caller (run-fft.c, read from its mounted
 * location) against include/offt.h, and compares struct layouts with the reference's offt.h; both need the FFTW type and
 * function NAMES those files mention to exist.  Nothing here is compiled into an object, linked, or shipped. */
#ifndef OFFT_TEST_SHIM_FFTW3_H
#define OFFT_TEST_SHIM_FFTW3_H
typedef double fftw_complex[2];
typedef struct offt_test_shim_fftw_plan_s *fftw_plan;
#ifndef FFTW_MEASURE
#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE (0U)
#define FFTW_EXHAUSTIVE (1U << 3)
#define FFTW_PATIENT (1U << 5)
#define FFTW_ESTIMATE (1U << 6)
#endif
void fftw_execute(const fftw_plan p);
void fftw_destroy_plan(fftw_plan p);
void fftw_print_plan(const fftw_plan p);
void *fftw_malloc(unsigned long n);
void fftw_free(void *p);
#endif
