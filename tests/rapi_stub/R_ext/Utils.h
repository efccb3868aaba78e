/* tests/rapi_stub/R_ext/Utils.h -- test stand-in: R_CheckUserInterrupt (Writing R Extensions 6.12) only counts its calls. */
#ifndef RAPI_STUB_UTILS_H
#define RAPI_STUB_UTILS_H
#ifdef __cplusplus
extern "C" {
#endif
void R_CheckUserInterrupt(void);
#ifdef __cplusplus
}
#endif
#endif
