// stand-in: see compat/mini_pcl.h
#pragma once
#include "../mini_pcl.h"
