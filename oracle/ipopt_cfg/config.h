/* Hand-written build configuration for compiling the UNMODIFIED reference
 * (coin-or/Ipopt 3.14.15, sources left in place under /root/reference/src)
 * into oracle/_ref/ without running its autotools build system.
 * TEST/ORACLE INFRASTRUCTURE ONLY - not part of the shipped product.
 * The macro names are the ones the reference's sources test for
 * (src/Common/config.h.in); the values describe this image
 * (gcc 11, glibc, LP64, LAPACK + PARDISO from oneMKL's libmkl_rt). */
#ifndef MI355X_REF_CONFIG_H
#define MI355X_REF_CONFIG_H
#define F77_FUNC(name,NAME) name ## _
#define F77_FUNC_(name,NAME) name ## _
#define HAVE_CFLOAT 1
#define HAVE_CMATH 1
#define HAVE_DLFCN_H 1
#define HAVE_FLOAT_H 1
#define HAVE_INTTYPES_H 1
#define HAVE_MATH_H 1
#define HAVE_STDINT_H 1
#define HAVE_STDIO_H 1
#define HAVE_STDLIB_H 1
#define HAVE_STRINGS_H 1
#define HAVE_STRING_H 1
#define HAVE_SYS_STAT_H 1
#define HAVE_SYS_TYPES_H 1
#define HAVE_UNISTD_H 1
#define HAVE_VSNPRINTF 1
#define STDC_HEADERS 1
#define SIZEOF_INT_P 8
#define IPOPT_CHECKLEVEL 0
#define IPOPT_VERBOSITY 0
#define IPOPT_C_FINITE std::isfinite
#define IPOPT_HAS_DRAND48 1
#define IPOPT_HAS_RAND 1
#define IPOPT_HAS_STD__RAND 1
#define IPOPT_HAS_VA_COPY 1
#define IPOPT_HAS_FEENABLEEXCEPT 1
#define IPOPT_HAS_LAPACK 1
#define IPOPT_HAS_PARDISO_MKL 1
#define IPOPT_HAS_LINEARSOLVERLOADER 1
#define IPOPT_LAPACK_FUNC(name,NAME) name ## _
#define IPOPT_LAPACK_FUNC_(name,NAME) name ## _
#define IPOPT_HSL_FUNC(name,NAME) name ## _
#define IPOPT_HSL_FUNC_(name,NAME) name ## _
#define IPOPT_WSMP_FUNC(name,NAME) name ## _
#define IPOPT_WSMP_FUNC_(name,NAME) name ## _
#define IPOPT_VERSION "3.14.15"
#define IPOPT_VERSION_MAJOR 3
#define IPOPT_VERSION_MINOR 14
#define IPOPT_VERSION_RELEASE 15
#define PACKAGE_NAME "Ipopt"
#define PACKAGE_STRING "Ipopt 3.14.15"
#define PACKAGE_VERSION "3.14.15"
#define IPOPTLIB_EXPORT __attribute__((__visibility__("default")))
#define IPOPTAMPLINTERFACELIB_EXPORT
#define SIPOPTLIB_EXPORT
#define HSLLIB_EXPORT
#endif
