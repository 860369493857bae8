// matching_scaling.h -- symmetric maximum-product-matching scaling (MC64-style), host side; see matching_scaling.cpp
#pragma once
namespace mi355x {
// ptr/idx/absval: FULL symmetric pattern by columns (both triangles), |values|; scale[n] out; returns false only on bad input
bool matching_scaling(int n, const int* ptr, const int* idx, const double* absval, double* scale, int* num_unmatched);
}
