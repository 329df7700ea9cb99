#ifndef RMOCK_RDYNLOAD_H_
#define RMOCK_RDYNLOAD_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_CMethodDef;
typedef R_CallMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo* info, const R_CMethodDef* c, const R_CallMethodDef* call,
                       const R_FortranMethodDef* f, const R_ExternalMethodDef* e);
int R_useDynamicSymbols(DllInfo* info, int value);
#ifdef __cplusplus
}
#endif
#endif
