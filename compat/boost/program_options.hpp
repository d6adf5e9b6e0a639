// stand-in: see compat/mini_boost_po.h
#pragma once
#include "../mini_boost_po.h"
