// stand-in: see compat/mini_pcl_io.h
#pragma once
#include "../../mini_pcl_io.h"
