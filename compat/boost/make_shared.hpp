// stand-in: boost::shared_ptr is std::shared_ptr (compat/mini_pcl.h)
#pragma once
#include "../mini_pcl.h"
