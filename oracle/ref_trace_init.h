// TEST INFRASTRUCTURE ONLY: force-included into the reference build (oracle/Makefile)
// so that the dormant per-level trace the reference prints to stderr
// (/root/reference/StrainCall/NonparametricClustering.cpp:287-298,460-471) can be
// emitted with more than the default 6 significant digits: SC_TRACE_PREC=<digits>.
#ifndef ORACLE_REF_TRACE_INIT_H
#define ORACLE_REF_TRACE_INIT_H
#include <cstdlib>
#include <iostream>
namespace {
struct OracleRefTraceInit {
    OracleRefTraceInit() {
        const char* p = std::getenv("SC_TRACE_PREC");
        if (p) std::cerr.precision(std::atoi(p));
    }
};
static OracleRefTraceInit oracle_ref_trace_init_instance;
}
#endif
