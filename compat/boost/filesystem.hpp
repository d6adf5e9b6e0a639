// stand-in: see compat/mini_boost_fs.h
#pragma once
#include "../mini_boost_fs.h"
