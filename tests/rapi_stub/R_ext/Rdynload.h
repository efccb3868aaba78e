/* tests/rapi_stub/R_ext/Rdynload.h -- test stand-in for the registration interface (Writing R Extensions 5.4): the
 * registered .Call table is kept so that the harness can look entry points up BY NAME with their arity, as .Call does. */
#ifndef RAPI_STUB_RDYNLOAD_H
#define RAPI_STUB_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_CMethodDef;
typedef R_CallMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct rapi_dllinfo { const R_CallMethodDef* call; int dynamic_symbols; } DllInfo;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* c, const R_CallMethodDef* call, const R_FortranMethodDef* f,
                       const R_ExternalMethodDef* ext);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
/* harness side: the registered routine `name` (NULL when absent), its arity in *nargs */
DL_FUNC rapi_lookup(const DllInfo* info, const char* name, int* nargs);
#ifdef __cplusplus
}
#endif
#endif
