/* Public-side configuration (what the reference would install as IpoptConfig.h's
 * payload) for code compiled AGAINST the reference headers: our adapter, the
 * ScalableProblems driver and the golden-dump harness.  Hand-written, see config.h. */
#ifndef MI355X_REF_CONFIG_IPOPT_H
#define MI355X_REF_CONFIG_IPOPT_H
#define IPOPT_VERSION "3.14.15"
#define IPOPT_VERSION_MAJOR 3
#define IPOPT_VERSION_MINOR 14
#define IPOPT_VERSION_RELEASE 15
#define IPOPT_CHECKLEVEL 0
#define IPOPT_VERBOSITY 0
#define IPOPTLIB_EXPORT
#define IPOPTAMPLINTERFACELIB_EXPORT
#define SIPOPTLIB_EXPORT
#define IPOPT_FORTRAN_INTEGER_TYPE ipindex
#endif
