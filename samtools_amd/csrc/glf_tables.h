// glf_tables.h -- layout of the error-model coefficient block (doubles): fk[256] | beta[64][256][256] | lhet[256][256]
#pragma once
#include <cstddef>
#include <vector>
namespace sta {
constexpr size_t GLF_FK_OFF = 0, GLF_BETA_OFF = 256, GLF_LHET_OFF = 256 + (size_t)64 * 256 * 256, GLF_TAB_DOUBLES = GLF_LHET_OFF + 256 * 256;
void glf_tables(double depcorr, std::vector<double> &t);
}
