// Stand-in for the reference's include/eigen_extensions/eigen_extensions.h, which pulls in Eigen/Sparse
// and boost::filesystem for (de)serialisers the TSDF library never calls.  Only the two ASCII routines
// TSDFVolumeOctree::save/load use are provided, restating eigen_extensions.h:249-294: header line
// "% rows cols", then the matrix through operator<< at precision 16.
#pragma once
#include <iostream>
#include <limits>
#include <sstream>
#include <string>

#include "../mini_eigen.h"

namespace eigen_extensions {

template <class S, int T, int U>
void serializeASCII(const Eigen::Matrix<S, T, U> &mat, std::ostream &strm) {
  const std::streamsize old_precision = strm.precision();
  strm.precision(16);
  strm << "% " << mat.rows() << " " << mat.cols() << std::endl;
  strm << mat << std::endl;
  strm.precision(old_precision);
}

template <class S, int T, int U>
void deserializeASCII(std::istream &strm, Eigen::Matrix<S, T, U> *mat) {
  std::string line;
  while (line.length() == 0) getline(strm, line);
  std::istringstream hdr(line.substr(1));
  int rows, cols;
  hdr >> rows;
  hdr >> cols;
  *mat = Eigen::Matrix<S, T, U>(rows, cols);
  for (int y = 0; y < rows; ++y) {
    getline(strm, line);
    std::istringstream iss(line);
    std::string token;
    for (int x = 0; x < cols; ++x) {
      iss >> token;
      if (token[0] == 'n') {
        mat->coeffRef(y, x) = std::numeric_limits<S>::quiet_NaN();
      } else {
        std::istringstream buf(token);
        buf >> mat->coeffRef(y, x);
      }
    }
  }
}

}  // namespace eigen_extensions
