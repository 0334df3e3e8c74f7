#pragma once
#include <opencv2/stub.h>
